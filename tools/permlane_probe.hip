// permlane_probe.hip -- what v_permlane32_swap_b32 does on gfx950 (used by spectrum_f64_fused.hip's
// store epilogue): r = __builtin_amdgcn_permlane32_swap(a, b, false, false).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o)
{
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned* d; unsigned h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("r[0]: lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("r[1]: lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
