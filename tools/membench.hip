// membench.hip -- what HBM rate does the fused kernel's ACCESS PATTERN reach
// with no arithmetic?  Per "frame": read 2 KiB, write 4 KiB (N=1024, K=1).
// Variants: load width (2-byte strided like the kernel, or 16-byte), nontemporal
// hints, waves per CU, frame->wave assignment (strided or contiguous chunks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int LOADW, bool NTL, bool NTS, bool CONTIG>
__global__ __launch_bounds__(64) void pat(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    const int t = threadIdx.x;
    const long nb = gridDim.x;
    long f0, f1, fs;
    if (CONTIG) { const long per = (nframes + nb - 1) / nb; f0 = blockIdx.x * per; f1 = f0 + per < nframes ? f0 + per : nframes; fs = 1; }
    else { f0 = blockIdx.x; f1 = nframes; fs = nb; }
    for (long f = f0; f < f1; f += fs) {
        float acc[16];
        if (LOADW == 2) {
            const uint16_t* src = reinterpret_cast<const uint16_t*>(in) + f * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned v = NTL ? __builtin_nontemporal_load(src + 64 * r + t) : src[64 * r + t];
                acc[r] = (float)(v & 0xff) + (float)(v >> 8);
            }
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4* src = reinterpret_cast<const u4*>(in) + f * 128;
            u4 a = NTL ? __builtin_nontemporal_load(src + t) : src[t];
            u4 b = NTL ? __builtin_nontemporal_load(src + 64 + t) : src[64 + t];
            acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; acc[3] = a.w; acc[4] = b.x; acc[5] = b.y; acc[6] = b.z; acc[7] = b.w;
#pragma unroll
            for (int r = 8; r < 16; ++r) acc[r] = acc[r - 8] * 0.5f;
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* dst = reinterpret_cast<f4*>(out + f * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 o = {acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
            if (NTS) __builtin_nontemporal_store(o, dst + 64 * s + t); else dst[64 * s + t] = o;
        }
    }
}

template <int LOADW, bool NTL, bool NTS, bool CONTIG>
void run(const char* name, int waves_per_cu, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_cu;
    for (int i = 0; i < 10; ++i) pat<LOADW, NTL, NTS, CONTIG><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipDeviceSynchronize());
    const int steps = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < steps; ++i) pat<LOADW, NTL, NTS, CONTIG><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / steps;
    printf("%-44s waves/CU %2d : %7.2f us  %6.0f GB/s\n", name, waves_per_cu, us, 6144.0 * nframes / us / 1e3);
    fflush(stdout);
}

int main()
{
    const long nframes = 65536;
    std::vector<uint8_t*> ins(4); std::vector<float*> outs(4);
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&ins[i], nframes * 2048)); CHECK(hipMalloc(&outs[i], nframes * 4096));
        CHECK(hipMemset(ins[i], 0x55 + i, nframes * 2048)); }
    for (int w : {8, 12, 16, 24, 32}) {
        run<2, false, false, false>("u16 loads, plain, strided frames", w, ins, outs, nframes);
        run<16, false, false, false>("16B loads, plain, strided frames", w, ins, outs, nframes);
        run<2, false, true, false>("u16 loads, NT stores, strided", w, ins, outs, nframes);
        run<2, true, true, false>("u16 loads, NT loads+stores, strided", w, ins, outs, nframes);
        run<16, true, true, false>("16B loads, NT loads+stores, strided", w, ins, outs, nframes);
        run<2, false, false, true>("u16 loads, plain, contiguous chunks", w, ins, outs, nframes);
        run<2, false, true, true>("u16 loads, NT stores, contiguous chunks", w, ins, outs, nframes);
    }
    return 0;
}
