// occ_check.hip -- do two 256-thread workgroups with 256 VGPRs and ~70 KiB of dynamic LDS each share a CU? (diagnostic)
// build: hipcc -O2 --offload-arch=gfx950 tools/occ_check.hip -o tools/build/occ_check
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, int spin)
{
    extern __shared__ double lds[];
    const unsigned long long t0 = wall_clock64();
    double acc = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");           // 256 VGPRs
    for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = acc; __syncthreads(); acc = acc * 1.0000001 + lds[(threadIdx.x + 1) & 255]; __syncthreads(); }
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = t0;
        out[3 * blockIdx.x + 1] = wall_clock64();
        out[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
    if (acc == 12345.678) out[0] = 0;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int lds_max = 0; CHECK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, 0));
    printf("CUs %d, max LDS per block %d, sharedMemPerMultiprocessor %zu\n", p.multiProcessorCount, lds_max, (size_t)p.maxSharedMemoryPerMultiProcessor);
    for (int lds : {32768, 65536, 69632, 71680, 73728, 81920}) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int nb = -1;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds));
        const int blocks = 2 * p.multiProcessorCount;
        unsigned long long* d; CHECK(hipMalloc(&d, blocks * 3 * sizeof(unsigned long long)));
        std::vector<unsigned long long> h(blocks * 3);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d, 20000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, h[3 * b]); t1 = std::max(t1, h[3 * b + 1]); }
        int late = 0; double dur = 0;
        for (int b = 0; b < blocks; ++b) { if ((h[3 * b] - t0) > (t1 - t0) / 4) ++late; dur += (double)(h[3 * b + 1] - h[3 * b]); }
        printf("LDS %6d B: occupancy API says %d blocks per CU; %d workgroups launched, %d of them started later than a quarter into the run; launch %.1f us, mean workgroup life %.1f us\n",
               lds, nb, blocks, late, (t1 - t0) / 100.0, dur / blocks / 100.0);
        CHECK(hipFree(d));
    }
    return 0;
}
