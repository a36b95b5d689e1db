// cicpat.hip -- what HBM rate does the CIC-fused kernel's ACCESS PATTERN reach with
// no arithmetic?  Per "frame" (N = 2048 outputs, 128 threads = 2 wavefronts): read
// 2048 * 2R bytes the way the input stage does, write 8 KiB the way the epilogue does
// (float2 per thread and p2).  Answers "how far below its own pattern's ceiling is
// cic8_2048pt / cic12_2048pt": tools/README.md.
//
//   cicpat <mode> <R> <G> <wgs_per_cu> [frames]
//     mode 0: 16-byte per-lane loads, all sixteen in flight (the R = 8 stage; R must be 8)
//     mode 1: LDS-DMA, 16 pieces of 128R bytes per wavefront in rounds of G (the R = 10 / 12 stage)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_row(float* out, long f, int t, const float (&acc)[16])
{
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        f2v o = {acc[2 * s], acc[2 * s + 1]};
        __builtin_nontemporal_store(o, reinterpret_cast<f2v*>(out + f * 2048 + s * 256) + t);
    }
}

__global__ __launch_bounds__(128) void pat_reg(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    const int t = threadIdx.x;
    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const u4* src = reinterpret_cast<const u4*>(in) + f * 2048;
        u4 raw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[r] = __builtin_nontemporal_load(src + 128 * r + t);
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (float)(raw[r].x + raw[r].y + raw[r].z + raw[r].w);
        store_row(out, f, t, acc);
    }
}

template <int R, int G>
__global__ __launch_bounds__(128) void pat_dma(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    constexpr int chunk = 128 * R, n16 = chunk >> 10, n4 = (chunk & 1023) >> 8;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    uint8_t* my = stage + w * (G * chunk);
    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const uint8_t* src = in + (f * 2048 + 64 * w) * R * 2;
        float acc[16];
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += G) {
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                const uint8_t* s = src + (long)(r0 + rr) * 128 * R * 2;
                uint8_t* d = my + rr * chunk;
#pragma unroll
                for (int i = 0; i < n16; ++i)
                    __builtin_amdgcn_global_load_lds((glb_vp)(s + i * 1024 + lane * 16), (lds_vp)(d + i * 1024), 16, 0, 2);
#pragma unroll
                for (int i = 0; i < n4; ++i)
                    __builtin_amdgcn_global_load_lds((glb_vp)(s + n16 * 1024 + i * 256 + lane * 4), (lds_vp)(d + n16 * 1024 + i * 256), 4, 0, 2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int rr = 0; rr < G; ++rr)
                acc[r0 + rr] = (float)*reinterpret_cast<const unsigned*>(my + rr * chunk + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        store_row(out, f, t, acc);
    }
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int R = argc > 2 ? atoi(argv[2]) : 8;
    const int G = argc > 3 ? atoi(argv[3]) : 4;
    const int wgs = argc > 4 ? atoi(argv[4]) : 8;
    long frames = argc > 5 ? atol(argv[5]) : (R == 8 ? 8192 : 5456);
    const int sets = 4, steps = 1500, settle = 500;
    const size_t in_b = (size_t)frames * 2048 * R * 2, out_b = (size_t)frames * 2048 * 4;
    std::vector<uint8_t*> ins(sets);
    std::vector<float*> outs(sets);
    for (int s = 0; s < sets; ++s) {
        CHECK(hipMalloc(&ins[s], in_b));
        CHECK(hipMalloc(&outs[s], out_b));
        CHECK(hipMemset(ins[s], 0x5a + s, in_b));
    }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const dim3 grid(256 * wgs), block(128);
    auto launch = [&](int s) {
        if (mode == 0) hipLaunchKernelGGL(pat_reg, grid, block, 0, st, ins[s], outs[s], frames);
        else if (R == 12 && G == 4) hipLaunchKernelGGL((pat_dma<12, 4>), grid, block, 2 * 4 * 1536, st, ins[s], outs[s], frames);
        else if (R == 12 && G == 8) hipLaunchKernelGGL((pat_dma<12, 8>), grid, block, 2 * 8 * 1536, st, ins[s], outs[s], frames);
        else if (R == 12 && G == 16) hipLaunchKernelGGL((pat_dma<12, 16>), grid, block, 2 * 16 * 1536, st, ins[s], outs[s], frames);
        else if (R == 8 && G == 4) hipLaunchKernelGGL((pat_dma<8, 4>), grid, block, 2 * 4 * 1024, st, ins[s], outs[s], frames);
        else if (R == 8 && G == 8) hipLaunchKernelGGL((pat_dma<8, 8>), grid, block, 2 * 8 * 1024, st, ins[s], outs[s], frames);
        else if (R == 8 && G == 16) hipLaunchKernelGGL((pat_dma<8, 16>), grid, block, 2 * 16 * 1024, st, ins[s], outs[s], frames);
        else { printf("unsupported mode/R/G\n"); exit(2); }
    };
    for (int i = 0; i < settle; ++i) launch(i % sets);
    CHECK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < steps; ++i) launch(i % sets);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / steps;
    printf("mode %d R %2d G %2d wgs/cu %2d frames %ld: %7.2f us  %6.0f GB/s (%.3f of 8 TB/s)\n", mode, R, G, wgs, frames,
           us, (double)(in_b + out_b) / us / 1e3, (double)(in_b + out_b) / us / 1e3 / 8000.0);
    return 0;
}
