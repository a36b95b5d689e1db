// spectrum_f64_4096y.hip -- 4096-point cmplx_u8 frames -> power spectra in DOUBLE for the windowed / K-frame rows
// (BASELINE configs[2] in the reference's arithmetic: src/spectrum.c:54-60 convert, :21 f64 forward DFT, :23-34
// |X|^2 + fft-shift + accumulate + DC-slot rule; K loop of src/cbb_main.c:50-59; dB / payload epilogue of
// src/cbb_main.c:121-130 in double; the Hann window is this build's extension).
//
// Why (round 6; profiles/r06_f64_4096_diag.txt): spectrum_f64_fused.hip at N = 4096 (16 x 16 x 16, four
// wavefronts per frame, TWO transpositions that both cross the wavefronts: four s_barrier per frame) ran the
// configs[2] rows in 218 us per 16 384 frames at 1 120 W -- under the power cap, at the full clock, i.e. waiting:
// 131 us without its LDS traffic and barriers, 181 with the barriers alone removed, and FASTER with one
// workgroup per CU (192 us) than with two.  This kernel keeps the decomposition and changes who talks to whom:
//
//   n = 256 r + 16 r2 + m2,   k = q1 + 16 q2 + 256 q3            (r, r2, m2, q1, q2, q3 < 16)
//   pass 1  thread t = 16 r2 + m2 of the frame's 256 loads x[256 r + t], r < 16 (one 128-byte line per
//           wave-instruction), converts, windows and transforms over r: plain radix-16, NO twiddle multiply.
//   exchange 1 -- the only one that crosses wavefronts -- (q1, t) -> row q1 of the buffer; wavefront w then
//           reads rows 4w .. 4w+3 only: its own quarter of the buffer.
//   pass 2  lane (q1, m2): radix-16 over r2 with the twiddle owed, W_4096^(t q1) = (W_256^q1)^r2 W_4096^(m2 q1),
//           absorbed as far as it is a geometric sequence in the register index (fft_regs_impl.h "last pass",
//           8 (cos, tan) pairs per lane); the lane-dependent rest is NOT applied here ...
//   exchange 2 stays inside the wavefront: (q1, m2; q2) -> lane (q2, q1), sixteen m2 contiguous, through the
//           wavefront's own quarter (rows of 16 padded to 17: the conflict-free image of spectrum_f64_1024x.hip):
//           no barrier, wavefront-scope fences only.
//   pass 3  radix-16 over m2.  What is owed now is W_4096^(m2 q1) W_256^(m2 q2) = (W_4096^(q1 + 16 q2))^m2 --
//           the rest of pass 1's twiddle AND pass 2's are one geometric sequence with a lane-constant ratio:
//           absorbed by the same fused-multiply-add form.  No complex twiddle multiplication anywhere:
//           148 + 192 + 192 = 532 f64 operations per 16 points instead of 148 + 62 + 148 + 192 = 550, and 64 VGPRs
//           of twiddles instead of 96 (the pass-3 pairs stay in registers: no LDS re-reads).
//   rows    lane (q2, q1) of wavefront w ends with bins k = q1 + 16 q2 + 256 q3, q3 over its sixteen registers
//           -- runs of 4 bins, 16 apart: once per ROW (K frames) the values go through the (then idle) buffer so
//           that every store instruction writes consecutive bytes.  Cheap for K >= 2, the rows this kernel is for.
//   samples  sixteen 2-byte loads per lane and frame cost this kernel a quarter of its time (the vector-memory
//           issue of 64 such instructions per frame, profiles/r06_f64_4096_diag.txt "noload"): every wavefront
//           instead copies ITS sixteen 128-byte pieces of the next frame into a 2 KiB LDS buffer of its own with
//           two global_load_lds_dwordx4 (LDS-DMA: no registers, two vector-memory instructions per wavefront
//           and frame) while the current frame is transformed, and reads them back with sixteen ds_read_u16.
//           Wavefront-private: no barrier; the copies are retired (vmcnt) at the top of the next frame.
// The 1/128 input scale rides on nothing: samples enter as (x - 128) w, the power sums are scaled by 2^-14 (exact)
// once per row.  Per frame: 2 barriers (around exchange 1), 64 LDS instructions per lane (72 before) + 16 two-byte reads, 632 f64
// operations per lane with twelve of the 16 Hann weights in registers and w_(r+8) = 1 - w_r for the rest (678 before).  Results equal spectrum_f64_fused.hip's
// to rounding; tests/test_f64_4096y_gpu.py holds both against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rtlws_internal.h"
#include "fft_regs_f64.h"

namespace rtlws {

using namespace f64;

constexpr int Y_ROW = 272;                       // exchange-1 row: 256 elements padded to 17 x 16 (double2 units)
constexpr int Y_ELEMS = 16 * Y_ROW;              // 4 352 double2 = 69 632 B; wavefront w's quarter: [w * 1088, (w + 1) * 1088)
constexpr int Y_RAW_BYTES = 4 * 2048;            // the wavefronts' raw-sample buffers, FIRST in the workgroup's LDS (M0 carries a 16-bit address)
constexpr size_t Y_LDS_BYTES = (size_t)Y_RAW_BYTES + (size_t)(Y_ELEMS + 1) * 16;     // + the DC hand-over slot: 77 840 B, two workgroups per CU

// This wavefront's sixteen 128-byte pieces of `frame` -- samples 256 r + 64 w .. + 63, r < 16 -- into its raw buffer as
// [r][64 samples]: lane L of copy j moves 16 bytes of piece r = 8 j + L / 8.  Inline asm on purpose (as in
// spectrum_fused.hip): hipcc would order every later LDS access of the workgroup's one shared array behind a
// __builtin_amdgcn_global_load_lds with vmcnt(0).  The ds_read_u16 of the CURRENT frame are waited for first
// (lgkmcnt(0)): the copies overwrite what they read.  M0 is saved and restored in the same statement.
// lane index rebuilt where it is needed (two instructions) instead of held in a register through the frame loop,
// where every register is taken: the allocator spilt such values
__device__ __forceinline__ int y_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ void y_dma_raw(const SpectraParamsF64& p, long frame, int w, unsigned lds_byte_addr)
{
    const int l = y_lane();
    const uint8_t* g0 = reinterpret_cast<const uint8_t*>(p.in) + frame * 8192 + (l >> 3) * 512 + w * 128 + (l & 7) * 16;
    const uint8_t* g1 = g0 + 4096;
    const unsigned d0 = __builtin_amdgcn_readfirstlane(lds_byte_addr), d1 = d0 + 1024;
    unsigned keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "s"(d0), "s"(d1) : "memory");
}

size_t spectra_f64_4096y_lds_bytes() { return Y_LDS_BYTES; }

template <bool WIN, int OUT, bool ROWF32>
__global__ __launch_bounds__(256, 2) void spectra_f64_4096y(const SpectraParamsF64 p)
{
    constexpr int N = 4096;
    static_assert(!(ROWF32 && OUT == OUT_PAYLOAD), "payload rows are bytes in either form");
    extern __shared__ __attribute__((aligned(16))) double2 lds_all[];
    double2* const ldsd = lds_all + Y_RAW_BYTES / 16;             // the exchange buffer, behind the raw-sample buffers

    const int tid = threadIdx.x;                  // t = 16 r2 + m2 in pass 1
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;      // (w: wave-uniform, a scalar register)
    const int K = p.k_avg;
    const long ngroups = p.ngroups;
    double2* const slice = ldsd + w * (4 * Y_ROW);               // this wavefront's quarter
    double* const dc_slot = reinterpret_cast<double*>(ldsd + Y_ELEMS);

    // this wavefront's raw-sample buffer (2 KiB) and its LDS byte address (for M0); the first frame's copy goes out
    // before anything else
    const uint16_t* const rawl = reinterpret_cast<const uint16_t*>(lds_all) + w * 1024 + l;
    const unsigned raw_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(lds_all) + w * 2048);
    if ((long)blockIdx.x < ngroups) y_dma_raw(p, (long)blockIdx.x * K, w, raw_addr);

    // lane constants, resident for the life of the (persistent) workgroup: pass 2's pairs of alpha = W_256^q1
    // (lane (q1, m2) = (tid >> 4, tid & 15)), pass 3's of beta = W_4096^(q1 + 16 q2) (lane (q2, q1): the host
    // lays the table out by thread), the Hann weights from two lane constants
    f2 twA[8], twB[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) twA[m] = p.tw2f[(tid >> 4) * 8 + m];
#pragma unroll
    for (int m = 0; m < 8; ++m) twB[m] = p.twyb[tid * 8 + m];
    const f2 wcs = WIN ? p.hann_csf[tid] : mk(0.0, 0.0);
    // (w_(r+8) = 1 - w_r: the last four weights are one subtraction per frame each instead of four register
    // pairs this kernel does not have -- with all sixteen resident the allocator spills)
    constexpr int WREG = 12;
    double win[WREG];
#pragma unroll
    for (int r = 0; r < WREG; ++r) win[r] = WIN ? hann_w(r, wcs) : 1.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(twA[m].x), "v"(twA[m].y), "v"(twB[m].x), "v"(twB[m].y));
    if constexpr (WIN) {
#pragma unroll
        for (int r = 0; r < WREG; ++r) asm volatile("" ::"v"(win[r]));
    }

    const int wp = l >> 4, wc = l & 15;           // exchange 2, writer side: lane (p, c) = (q1 & 3, m2)
#ifdef RTLWS_Y_STAMP      // diagnostic build (tools/r6_phase_times.py): shader clocks per phase, summed over the frames of a wavefront
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64(), nfr = 0;
    const unsigned long long t_begin = wall_clock64();
#ifndef RTLWS_Y_STAMPS
#define RTLWS_Y_STAMPS 255      // which phase boundaries are stamped (bit i = boundary i)
#endif
#define Y_PHASE(i) do { if constexpr ((RTLWS_Y_STAMPS >> (i)) & 1) { const unsigned long long tn = clock64(); const long d_ = (long)(tn - tlast); if (d_ > 0) ph[i] += (unsigned long long)d_; tlast = tn; } } while (0)
#define Y_PIN_V() do { _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) asm volatile("" : "+v"(v[s_].x), "+v"(v[s_].y)); } while (0)
#elif defined(RTLWS_Y_BLOCKY)     // the stamp build's phase structure without the stamps: mask bit i keeps boundary i
#define Y_PHASE(i) do { } while (0)
#define Y_PIN_V() do { _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) asm volatile("" : "+v"(v[s_].x), "+v"(v[s_].y)); } while (0)
#else
#define Y_PHASE(i) do { } while (0)
#define Y_PIN_V() do { } while (0)
#endif
#ifndef RTLWS_Y_PINS
#define RTLWS_Y_PINS 7
#endif
#define Y_PIN_IF(bit) do { if constexpr ((RTLWS_Y_PINS >> bit) & 1) Y_PIN_V(); } while (0)

    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        double acc[16];
        double wdc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;

        for (int kf = 0; kf < K; ++kf) {
            const long frame = g * K + kf;
            f2 v[16];
            // this frame's samples have landed in the wavefront's buffer (its copies are the only vector-memory
            // loads in flight; the previous row's stores are retired with them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            Y_PHASE(7);       // wait for the samples' copy
            unsigned raw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) raw[r] = rawl[64 * r];
            {
                long nf = frame + 1;
                if (kf + 1 == K) nf = (g + gridDim.x) * K;
                if (nf < ngroups * K) y_dma_raw(p, nf, w, raw_addr);       // (wave-uniform; waits for the reads above)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // rectangular: (double)u8, the 128 offset kept (it only reaches bin 0, which is never output:
                // src/spectrum.c:31); windowed: (x - 128) w with the subtraction on the integers (exact) -- one
                // rounding, the same value as fma(x, w, -128 w), without sixteen more lane constants
                if constexpr (WIN) {
                    const int re = (int)(raw[r] & 0xffu) - 128, im = (int)((raw[r] >> 8) & 0xffu) - 128;
                    const double wr = r < WREG ? win[r] : 1.0 - win[r - 8];
                    v[r] = mk((double)re * wr, (double)im * wr);
                } else {
                    v[r] = mk((double)(raw[r] & 0xffu), (double)((raw[r] >> 8) & 0xffu));
                }
            }

            // ---- pass 1: radix-16 over r; slot s holds q1 = rev16(s)
            fft16_sel(v);
            Y_PIN_IF(0);
            Y_PHASE(0);       // read the samples, issue the next copy, convert, window, pass 1

            // ---- exchange 1 (all four wavefronts): (q1, t) -> row q1
            __syncthreads();            // the quarters are free: every wavefront is through its pass-3 reads
            Y_PHASE(1);       // barrier 1
#pragma unroll
            for (int s = 0; s < 16; ++s) ldsd[rev16(s) * Y_ROW + tid] = v[s];
            __syncthreads();
            Y_PHASE(2);       // exchange-1 writes + barrier 2
            // lane (q1, m2) = (tid >> 4, tid & 15) reads r2 = 0 .. 15: rows 4w .. 4w+3 only
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) v[r2] = ldsd[(tid >> 4) * Y_ROW + 16 * r2 + (tid & 15)];

            // ---- pass 2: radix-16 over r2, (W_256^q1)^r2 absorbed; slot s holds q2 = rev16(s)
            fft_last<16>(v, 0, twA);
            Y_PIN_IF(1);
            Y_PHASE(3);       // exchange-1 reads + pass 2

            // ---- exchange 2, inside the wavefront and inside its own quarter (which only this wavefront reads):
            // (p, c; q2) -> lane 4 q2 + p, sixteen c contiguous.  LDS operations of one wavefront execute in
            // order: the writes cannot pass the reads above.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) slice[17 * (4 * rev16(s) + wp) + wc] = v[s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = slice[17 * l + c];
            Y_PIN_IF(2);
            Y_PHASE(4);       // exchange 2 (writes, reads arrived)

            // ---- pass 3: radix-16 over m2, (W_4096^(q1 + 16 q2))^m2 absorbed; slot s holds q3 = rev16(s)
            fft_last<16>(v, 0, twB);

            // ---- |X|^2, accumulate; bin N-1 (q1 = q2 = q3 = 15: wavefront 3, lane 63, slot 15) also feeds the
            // DC slot with weight K - kf
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u == 15) {
                    const double pw = fma(v[u].y, v[u].y, v[u].x * v[u].x);
                    acc[u] += pw;
                    wdc = fma((double)(K - kf), pw, wdc);
                } else {
                    acc[u] = fma(v[u].y, v[u].y, fma(v[u].x, v[u].x, acc[u]));
                }
            }
#if defined(RTLWS_Y_STAMP) || defined(RTLWS_Y_BLOCKY)
            if constexpr ((RTLWS_Y_PINS >> 3) & 1) {
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(acc[u]));
            }
#endif
#ifdef RTLWS_Y_STAMP
            ++nfr;
#endif
            Y_PHASE(5);       // pass 3 + |X|^2
        }

        // ---- row: DC-slot rule (src/spectrum.c:25-33: slot N/2 -- bin 0: wavefront 0, lane 0, slot 0 -- takes
        // sum_k (K-k) P_k[N-1]), epilogue, and the trip through the buffer that makes the stores consecutive
        // (the row's addresses are rebuilt from the lane index: hoisted out of the row loop they would be live
        // through the frame loop)
        const int tr = 64 * w + y_lane();
        if (tr == 255) *dc_slot = wdc;
        __syncthreads();                // every wavefront is through its last pass-3 reads; the DC value is there
        if (tr == 0) acc[0] = *dc_slot;
        {
            // lane (q2, q1) = (l >> 2, 4 w + (l & 3)) holds bins k = q1 + 16 q2 + 256 q3; fft-shift = q3 ^ 8
            const int kk = 4 * (tr >> 6) + (tr & 3) + 16 * ((tr & 63) >> 2);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = kk + 256 * (rev16(u) ^ 8);
                const double a = acc[u] * 0x1p-14;              // the (x - 128) scale: (1/128)^2, exact
                if constexpr (OUT == OUT_PAYLOAD) {
                    // src/cbb_main.c:125-128, same operation order, in double
                    const double d = 10.0 * log10(fabs(p.lin_gain * a / (double)p.count));
                    const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
                    reinterpret_cast<uint8_t*>(ldsd)[i] = (uint8_t)m;
                } else {
                    const double o = (OUT == OUT_DB) ? 10.0 * log10(a / (double)p.count) : a;
                    if constexpr (ROWF32) reinterpret_cast<float*>(ldsd)[i] = (float)o;
                    else reinterpret_cast<double*>(ldsd)[i] = o;
                }
            }
        }
        __syncthreads();
        if constexpr (OUT == OUT_PAYLOAD) {
            // 16 consecutive bytes per thread: one 16-byte store, 4 KiB of consecutive bytes per workgroup
            typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
            const nt_u4 b = reinterpret_cast<const nt_u4*>(ldsd)[tr];
            __builtin_nontemporal_store(b, reinterpret_cast<nt_u4*>(reinterpret_cast<uint8_t*>(p.out) + g * N) + tr);
        } else if constexpr (ROWF32) {
            float* dst = reinterpret_cast<float*>(p.out) + g * N;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_nontemporal_store(reinterpret_cast<const float*>(ldsd)[256 * j + tr], dst + 256 * j + tr);
        } else {
            double* dst = reinterpret_cast<double*>(p.out) + g * N;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_nontemporal_store(reinterpret_cast<const double*>(ldsd)[256 * j + tr], dst + 256 * j + tr);
        }
        // (the next frame's first barrier orders these reads before the next exchange-1 writes)
        Y_PHASE(6);           // row epilogue
    }
#ifdef RTLWS_Y_STAMP
    // per wavefront: {phase sums 0..7, frames, start, end (100 MHz), magic} as twelve u64 BEHIND the last row of the
    // output buffer (tools/r6_phase_times.py allocates the room)
    if (l == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* st = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.out) + ngroups * N * (ROWF32 ? 4 : 8)) + (long)(blockIdx.x * 4 + w) * 12;
        for (int i = 0; i < 8; ++i) st[i] = ph[i];
        st[8] = nfr;
        st[9] = t_begin;
        st[10] = wall_clock64();
        st[11] = 0x5354414d50ull;
    }
#endif
}

template <bool WIN, int OUT, bool ROWF32>
static hipError_t launch_y_one(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    static std::atomic<unsigned long long> ready{0};          // > 64 KiB of LDS: the attribute once per device
    const unsigned long long bit = 1ull << (device & 63);
    if (!(ready.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectra_f64_4096y<WIN, OUT, ROWF32>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)Y_LDS_BYTES);
        if (e != hipSuccess) return e;
        ready.fetch_or(bit, std::memory_order_release);
    }
    if (blocks <= 0) return hipSuccess;      // rtlws_engine_prepare_f64: the attribute only, nothing enqueued
    hipLaunchKernelGGL((spectra_f64_4096y<WIN, OUT, ROWF32>), dim3(blocks), dim3(256), Y_LDS_BYTES, st, p);
    return hipGetLastError();
}

template <bool WIN>
static hipError_t launch_y_o(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    switch (p.out_mode) {
    case OUT_SUM: return p.rows_f32 ? launch_y_one<WIN, OUT_SUM, true>(p, blocks, st, device)
                                    : launch_y_one<WIN, OUT_SUM, false>(p, blocks, st, device);
    case OUT_DB: return p.rows_f32 ? launch_y_one<WIN, OUT_DB, true>(p, blocks, st, device)
                                   : launch_y_one<WIN, OUT_DB, false>(p, blocks, st, device);
    default: return launch_y_one<WIN, OUT_PAYLOAD, false>(p, blocks, st, device);
    }
}

// cmplx_u8 frames of 4096 points, any K >= 1, Hann or rectangular; `blocks` persistent workgroups of 256 threads
hipError_t launch_spectra_f64_4096y(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    return p.window ? launch_y_o<true>(p, blocks, st, device) : launch_y_o<false>(p, blocks, st, device);
}

}  // namespace rtlws
