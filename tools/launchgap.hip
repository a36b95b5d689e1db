// launchgap.hip -- fixed cost of one launch of a grid shaped like the fused
// kernel's (4096 single-wavefront workgroups, 9 KiB of LDS each), back to back on
// one stream: empty kernel, and a kernel that only loads the per-thread tables.
// Build: hipcc -O3 --offload-arch=gfx950 tools/launchgap.hip -o tools/build/launchgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(64, 4) void k_empty(float* out)
{
    extern __shared__ float lds[];
    if (out == nullptr) { lds[threadIdx.x] = 1.0f; out[0] = lds[63 - threadIdx.x]; }
}

__global__ __launch_bounds__(64, 4) void k_tables(const float2* __restrict__ tw, float* out)
{
    extern __shared__ float lds[];
    float2 t[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) t[s] = tw[threadIdx.x * 16 + s];
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc += t[s].x * t[s].y;
    if (acc == 123.456f) { lds[threadIdx.x] = acc; out[0] = lds[63 - threadIdx.x]; }
}

int main()
{
    float2* tw; float* out;
    CHECK(hipMalloc(&tw, 64 * 16 * sizeof(float2))); CHECK(hipMemset(tw, 0, 64 * 16 * sizeof(float2)));
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int steps = 2000;
    for (int grid : {256, 1024, 4096, 16384}) {
        for (int lds : {0, 9232}) {
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), lds, 0, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), lds, 0, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty   grid %5d lds %5d : %6.2f us per launch\n", grid, lds, 1e3 * ms / steps);
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k_tables, dim3(grid), dim3(64), lds, 0, tw, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("tables  grid %5d lds %5d : %6.2f us per launch\n", grid, lds, 1e3 * ms / steps);
        }
    }
    return 0;
}
