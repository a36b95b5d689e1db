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

// frame -> wave mappings: 0 strided (f = b + i*nb), 1 each XCD (b % 8) sweeps its own
// contiguous eighth of the batch, 2 blocks of 8 consecutive frames rotate over XCDs
template <int MAP>
__global__ __launch_bounds__(64) void pat_map(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    const long nb = gridDim.x, b = blockIdx.x;
    const long iters = nframes / nb;
    for (long i = 0; i < iters; ++i) {
        long f;
        if (MAP == 0) f = b + i * nb;
        else if (MAP == 1) f = (b % 8) * (nframes / 8) + (b / 8) + i * (nb / 8);
        else f = ((b / 8) + i * (nb / 8)) * 8 + (b % 8);       // == MAP 0 reordered: sanity
        float acc[16];
        const uint16_t* src = reinterpret_cast<const uint16_t*>(in) + f * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned v = __builtin_nontemporal_load(src + 64 * r + t);
            acc[r] = (float)(v & 0xff) + (float)(v >> 8);
        }
        f4* dst = reinterpret_cast<f4*>(out + f * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 o = {acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
            __builtin_nontemporal_store(o, dst + 64 * s + t);
        }
    }
}

template <int MAP>
void run_map(int waves_per_cu, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_cu;
    for (int i = 0; i < 10; ++i) pat_map<MAP><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipDeviceSynchronize());
    const int steps = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < steps; ++i) pat_map<MAP><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / steps;
    printf("frame map %d, NT                                waves/CU %2d : %7.2f us  %6.0f GB/s\n", MAP, waves_per_cu, us, 6144.0 * nframes / us / 1e3);
    fflush(stdout);
}

// access pattern of a 32x32 decomposition with two frames per wavefront: lane (h, m) loads
// x[32*r + m] of frame f+h (two 64-byte pieces per load instruction) and stores bin 32*p + m
// (two 128-byte pieces per dword store instruction).
__global__ __launch_bounds__(64) void pat_r32(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    const int t = threadIdx.x, h = t >> 5, m = t & 31;
    for (long f0 = (long)blockIdx.x * 2; f0 < nframes; f0 += (long)gridDim.x * 2) {
        const long f = f0 + h;
        const uint16_t* src = reinterpret_cast<const uint16_t*>(in) + f * 1024;
        float acc[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            unsigned v = __builtin_nontemporal_load(src + 32 * r + m);
            acc[r] = (float)(v & 0xff) + (float)(v >> 8);
        }
        float* dst = out + f * 1024;
#pragma unroll
        for (int p = 0; p < 32; ++p) __builtin_nontemporal_store(acc[p], dst + 32 * p + m);
    }
}

void run_r32(int waves_per_cu, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_cu;
    for (int i = 0; i < 10; ++i) pat_r32<<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipDeviceSynchronize());
    const int steps = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < steps; ++i) pat_r32<<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / steps;
    printf("32x32 pattern, 2 frames per wave, NT              waves/CU %2d : %7.2f us  %6.0f GB/s\n", waves_per_cu, us, 6144.0 * nframes / us / 1e3);
    fflush(stdout);
}

// FPI consecutive frames per wave-iteration (bigger contiguous bursts per wave); NT both
template <int FPI>
__global__ __launch_bounds__(64) void pat_burst(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    for (long f0 = (long)blockIdx.x * FPI; f0 < nframes; f0 += (long)gridDim.x * FPI) {
        float acc[FPI][16];
#pragma unroll
        for (int j = 0; j < FPI; ++j) {
            const uint16_t* src = reinterpret_cast<const uint16_t*>(in) + (f0 + j) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned v = __builtin_nontemporal_load(src + 64 * r + t);
                acc[j][r] = (float)(v & 0xff) + (float)(v >> 8);
            }
        }
#pragma unroll
        for (int j = 0; j < FPI; ++j) {
            f4* dst = reinterpret_cast<f4*>(out + (f0 + j) * 1024);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f4 o = {acc[j][4 * s], acc[j][4 * s + 1], acc[j][4 * s + 2], acc[j][4 * s + 3]};
                __builtin_nontemporal_store(o, dst + 64 * s + t);
            }
        }
    }
}

template <int FPI>
void run_burst(int waves_per_cu, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * waves_per_cu;
    for (int i = 0; i < 10; ++i) pat_burst<FPI><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipDeviceSynchronize());
    const int steps = 200;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < steps; ++i) pat_burst<FPI><<<blocks, 64>>>(ins[i % ins.size()], outs[i % outs.size()], nframes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / steps;
    printf("burst of %d frames per wave-iteration, NT       waves/CU %2d : %7.2f us  %6.0f GB/s\n", FPI, waves_per_cu, us, 6144.0 * nframes / us / 1e3);
    fflush(stdout);
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

// one-direction and 1:1 ceilings with the same wave-per-2KiB-chunk structure
template <int MODE, bool NT>   // 0 read-only (sum), 1 write-only, 2 copy 1:1 (4 KiB in, 4 KiB out)
__global__ __launch_bounds__(64) void stream_k(const float* __restrict__ in, float* __restrict__ out, long nchunks)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {     // chunk = 4 KiB
        const f4* src = reinterpret_cast<const f4*>(in) + c * 256;
        f4* dst = reinterpret_cast<f4*>(out) + c * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 v = {1.0f, 2.0f, 3.0f, (float)c};
            if (MODE != 1) v = NT ? __builtin_nontemporal_load(src + 64 * s + t) : src[64 * s + t];
            if (MODE == 0) acc += v;
            else if (NT) __builtin_nontemporal_store(v, dst + 64 * s + t);
            else dst[64 * s + t] = v;
        }
    }
    if (MODE == 0 && acc.x == 123.456f) out[t] = acc.y;
}

template <int MODE, bool NT>
void run_stream(const char* name, float* a, float* b, long bytes_each)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long nchunks = bytes_each / 4096;
    for (int w : {8, 16, 32}) {
        const int blocks = 256 * w;
        stream_k<MODE, NT><<<blocks, 64>>>(a, b, nchunks);
        CHECK(hipDeviceSynchronize());
        const int steps = 100;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < steps; ++i) stream_k<MODE, NT><<<blocks, 64>>>(a, b, nchunks);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / steps;
        const double moved = (MODE == 2 ? 2.0 : 1.0) * bytes_each;
        printf("%-30s waves/CU %2d : %8.2f us  %6.0f GB/s\n", name, w, us, moved / us / 1e3);
    }
    fflush(stdout);
}

int main()
{
    {
        const long bytes = 1L << 30;      // 1 GiB each: far past the 256 MiB Infinity Cache
        float *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
        run_stream<0, false>("read-only 1 GiB", a, b, bytes);
        run_stream<0, true>("read-only 1 GiB NT", a, b, bytes);
        run_stream<1, false>("write-only 1 GiB", a, b, bytes);
        run_stream<1, true>("write-only 1 GiB NT", a, b, bytes);
        run_stream<2, false>("copy 1 GiB -> 1 GiB", a, b, bytes);
        run_stream<2, true>("copy 1 GiB -> 1 GiB NT", a, b, bytes);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    const long nframes = 65536;
    std::vector<uint8_t*> ins(4); std::vector<float*> outs(4);
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&ins[i], nframes * 2048)); CHECK(hipMalloc(&outs[i], nframes * 4096));
        CHECK(hipMemset(ins[i], 0x55 + i, nframes * 2048)); }
    for (int w : {4, 8, 12, 16}) run_r32(w, ins, outs, nframes);
    for (int w : {8, 16}) { run_map<0>(w, ins, outs, nframes); run_map<1>(w, ins, outs, nframes); run_map<2>(w, ins, outs, nframes); }
    for (int w : {4, 8, 16}) { run_burst<1>(w, ins, outs, nframes); run_burst<2>(w, ins, outs, nframes); run_burst<4>(w, ins, outs, nframes); }
    for (int w : {8, 16}) {
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
