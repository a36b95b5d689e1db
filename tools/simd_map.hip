// simd_map.hip -- which SIMD does wavefront i of a workgroup land on? (diagnostic for the 8-wavefront workgroups)
// build: hipcc -O2 --offload-arch=gfx950 tools/simd_map.hip -o tools/build/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int T>
__global__ __launch_bounds__(T, 2) void k(unsigned* out)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = 1.0;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (T / 64) + threadIdx.x / 64] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}
template <int T>
int run(int lds)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256, W = T / 64;
    unsigned* d; CHECK(hipMalloc(&d, blocks * W * sizeof(unsigned)));
    std::vector<unsigned> h(blocks * W);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(T), lds, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < W; ++w) hist[w][(h[b * W + w] >> 4) & 3]++;
    printf("%d threads, %d B of LDS: SIMD of wavefront i, over %d workgroups\n", T, lds, blocks);
    for (int w = 0; w < W; ++w) printf("  wavefront %d: SIMD0 %3d  SIMD1 %3d  SIMD2 %3d  SIMD3 %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  first workgroups:");
    for (int b = 0; b < 4; ++b) { printf(" ["); for (int w = 0; w < W; ++w) printf("%u", (h[b * W + w] >> 4) & 3); printf("]"); }
    printf("\n");
    CHECK(hipFree(d));
    return 0;
}
int main() { return run<256>(77840) || run<512>(155680) || run<512>(32768); }
