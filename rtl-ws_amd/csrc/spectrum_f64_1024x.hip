// spectrum_f64_1024x.hip -- 1024-point rectangular cmplx_u8 frames -> power spectra in DOUBLE with
// ONE LDS transposition per frame instead of two (the reference's arithmetic: src/spectrum.c:54-60
// convert, :21 f64 forward DFT, :23-34 |X|^2 + fft-shift + accumulate + DC-slot rule; K loop of
// src/cbb_main.c:50-59; dB / payload epilogue of src/cbb_main.c:121-130 in double).
//
// Why: spectrum_f64_fused.hip (1024 = 16 x 16 x 4, two transpositions of 16-byte elements) moves
// 64 KiB per frame through LDS, and on gfx950 a ds_write_b128 costs 13 cycles per wave-instruction
// (~79 B/clk/CU, a third of the read rate): with f32 rows the kernel is bound by that and by the
// f64 issue rate together, not by HBM (ablations: 118 us per 65 536 frames, 90 us without the LDS
// traffic, stores free; profiles/r04_ab_f64_variants.txt).  This kernel decomposes
//      1024 = 4 x 256,   256 = 16 x 16
// and does the radix-4 FIRST, where the data are still the 8-bit samples:
//   pass 0  n = 256 a + m: lane t loads x[64 j + t], j < 16 (one 128-byte line per wave-instruction,
//           as everywhere), i.e. a = j / 4 and m = 64 b + t with b = j % 4.  y_p[m] = sum_a x[256a+m]
//           (-i)^(a p) is exact INTEGER arithmetic on packed int16 (re, im) pairs (|y| <= 1020):
//           v_pk_add_i16 / v_pk_sub_i16, 11 instructions per b.
//   cross-row transpose  the 256-point transform p wants m = c + 16 r on lane (p, c): the 4 x 4
//           exchange (lane row t / 16  <->  index p) of ONE dword per value is two
//           v_permlane16_swap + two v_permlane32_swap per b -- 16 VALU instructions per frame, no LDS.
//           Lane (row p, column c) then holds y_p[c + 16 r], r = 4 b + g, r = 0 .. 15 in order.
//   pass A  the twiddle owed, W_1024^(p m) = W_1024^(p c) (W_64^p)^r, is a lane constant times a
//           geometric sequence in the register index: the radix-16 over r absorbs the sequence in
//           fused-multiply-add form (fft_regs_impl.h "last pass": 8 (cos, tan) pairs per lane, 192
//           operations), the constant rides on ...
//   twB     ... the inner twiddles W_256^(c q) (one complex multiply per point, which also carries the
//           exact 1/128 input scale), then the ONE LDS transposition (16-byte elements, 32 KiB of LDS
//           traffic per frame), lane t = 4 q + p reading its sixteen c contiguously,
//   pass B  radix-16 over c: lane t ends with bins k = 64 q' + t, q' = 0 .. 15 -- for every q' the
//           wavefront stores 64 consecutive outputs (256 B of f32 / 512 B of f64), lane-contiguous.
// f64 operations per frame: 192 + 64 + 148 + 32 (|X|^2) = 436 against 484, LDS bytes halved, no
// workgroup barrier (one wavefront per frame).  Same results as spectrum_f64_fused.hip to rounding
// (strict-metric error ~1e-12); tests/test_f64_1024x_gpu.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rtlws_internal.h"
#include "fft_regs_f64.h"

#define RTLWS_X_LDS_ORDER 0

namespace rtlws {

using namespace f64;

typedef short pk_i16 __attribute__((ext_vector_type(2)));      // (re, im) of one integer point

// V_PERMLANE16_SWAP: odd rows (16 lanes) of a <-> even rows of b; V_PERMLANE32_SWAP: upper half of a
// <-> lower half of b (tools/permlane_probe.hip)
__device__ __forceinline__ void swap_rows16(unsigned& a, unsigned& b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap_rows32(unsigned& a, unsigned& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// WAVES = 1: one wavefront per workgroup, output rows dealt statically (row = workgroup + i * grid) -- small
// batches and the drop-in calls.  WAVES > 1: ONE workgroup of WAVES wavefronts per CU (WAVES / 4 per SIMD),
// the workgroup's rows (b + i * grid) handed to its wavefronts one at a time through a counter in LDS.  Why
// (profiles/r05_wave_timeline_2_per_simd.txt): the SIMD arbitrates its vector pipe by age, the older of two
// co-resident wavefronts runs at 0.95 of its solo rate and the younger gets the leftover slots -- with rows dealt
// statically the older one left after 122 of 198 k cycles and the younger finished its last 21 rows alone, at
// the solo rate (3 640 cycles per row against 2 820 for the pair).  Nothing but the row-to-wavefront map differs:
// no barrier in the frame loop (the wavefronts share no LDS data), results bit-identical.
template <int OUT, bool KONE, bool ROWF32, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 1 ? 2 : WAVES / 4) void spectra_f64_1024x(const SpectraParamsF64 p)
{
    constexpr int N = 1024;
    static_assert(!(ROWF32 && OUT == OUT_PAYLOAD), "payload rows are bytes in either form");
    static_assert(WAVES == 1 || WAVES % 4 == 0, "whole wavefronts per SIMD");
    extern __shared__ __attribute__((aligned(16))) double2 lds_all[];
    // (three wavefronts per SIMD -- WAVES = 12 at 164 VGPRs, inner twiddles and a two-halves transposition in LDS --
    // was built and measured slower, profiles/r05_ab_three_wavefronts_per_simd.txt; tools/variants/csrc_hooks.patch)
    constexpr int SLICE_F2 = 17 * 64;                                  // double2 elements per wavefront
    double2* const ldsd = lds_all + (WAVES == 1 ? 0 : (threadIdx.x >> 6) * SLICE_F2);   // this wavefront's own slice
    unsigned* const row_counter = reinterpret_cast<unsigned*>(lds_all + WAVES * SLICE_F2);    // (WAVES > 1)

    const int t = threadIdx.x & 63;
    const int K = KONE ? 1 : p.k_avg;
    const long ngroups = p.ngroups;
    // this workgroup's output rows: blockIdx.x + i * gridDim.x; the first WAVES indices i are dealt statically
    long g = (long)blockIdx.x + (WAVES == 1 ? 0L : (long)(threadIdx.x >> 6) * gridDim.x);
    if constexpr (WAVES > 1) {
        if (threadIdx.x == 0) *row_counter = WAVES;
        __syncthreads();                                        // the only barrier of the kernel
    }
    // the row after `cur`: WAVES = 1 strides, WAVES > 1 takes the workgroup's next undone row (lane 0's LDS
    // atomic, broadcast); >= ngroups: none left
    auto next_row = [&](long cur) -> long {
        if constexpr (WAVES == 1) {
            return cur + gridDim.x;
        } else {
            unsigned v = 0;
            if (t == 0) v = __hip_atomic_fetch_add(row_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return (long)blockIdx.x + (long)(unsigned)__builtin_amdgcn_readfirstlane((int)v) * gridDim.x;
        }
    };

    unsigned raw[16];
    auto load_raw = [&](long frame) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(p.in) + frame * N;
#pragma unroll
        for (int j = 0; j < 16; ++j) raw[j] = __builtin_nontemporal_load(src + 64 * j + t);
    };
    if (g < ngroups) load_raw(g * K);

    // lane constants, resident for the life of the (persistent) workgroup
    f2 twA[8], twB[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) twA[m] = p.twxa[(t >> 4) * 8 + m];
#pragma unroll
    for (int s = 0; s < 16; ++s) twB[s] = p.twxb[t * 16 + s];
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(twA[m].x), "v"(twA[m].y));      // retired before the loop
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(twB[s].x), "v"(twB[s].y));

    const int wp = t >> 4, wc = t & 15;         // writer side of the transposition: lane (p, c)

    while (g < ngroups) {
        const long g_next = next_row(g);
        double acc[16];
        double wdc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;

        for (int kf = 0; kf < K; ++kf) {
            const long frame = g * K + kf;

            // ---- pass 0: radix-4 over a on packed int16 points.  The 128 offset of the samples
            // only reaches y_0 -> bin 0 of every 256-point transform p = 0 -> bins k = 4 k' + 0 ...
            // no: it reaches exactly X[0] (a constant sequence has a single non-zero bin), which is
            // never output (src/spectrum.c:31); it is kept, as in the other kernels.
            unsigned y[16];                     // y[4 b + pp]
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                pk_i16 x[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const unsigned r = raw[4 * a + b];
                    // bytes (re, im) -> int16 pair (re | im << 16): one v_perm_b32
                    x[a] = __builtin_bit_cast(pk_i16, __builtin_amdgcn_perm(r, r, 0x0c010c00u));
                }
                const pk_i16 s0 = x[0] + x[2], s1 = x[0] - x[2], s2 = x[1] + x[3], s3 = x[1] - x[3];
                const pk_i16 rot = {s3.y, (short)-s3.x};                 // -i * s3
                y[4 * b + 0] = __builtin_bit_cast(unsigned, (pk_i16)(s0 + s2));
                y[4 * b + 1] = __builtin_bit_cast(unsigned, (pk_i16)(s1 + rot));
                y[4 * b + 2] = __builtin_bit_cast(unsigned, (pk_i16)(s0 - s2));
                y[4 * b + 3] = __builtin_bit_cast(unsigned, (pk_i16)(s1 - rot));
            }
            {
                long nf = frame + 1;
                if (kf + 1 == K) nf = g_next * K;
                if (nf >= ngroups * K) nf = frame;        // in bounds, result unused  (this form: the ternary
                                                          // spelling costs the K > 1 instantiations 12-15 spilled VGPRs)
                load_raw(nf);
            }

            // ---- cross-row 4 x 4 transpose (lane row <-> p): afterwards lane (row p, column c)
            // holds y_p[c + 16 (4 b + g)] in y[4 b + g]
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                swap_rows16(y[4 * b + 0], y[4 * b + 1]);
                swap_rows16(y[4 * b + 2], y[4 * b + 3]);
                swap_rows32(y[4 * b + 0], y[4 * b + 2]);
                swap_rows32(y[4 * b + 1], y[4 * b + 3]);
            }
            f2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = mk((double)(short)(y[r] & 0xffffu), (double)((int)y[r] >> 16));

            // ---- pass A: radix-16 over r with the geometric pre-twiddle (W_64^p)^r absorbed;
            // slot s holds index q = rev16(s)
            fft_last<16>(v, 0, twA);
            // inner twiddles W_256^(c q) x the lane constant W_1024^(p c) x 1/128
#pragma unroll
            for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], twB[s]);

            // ---- the one transposition: (p, c; q) -> lane 4 q + p, sixteen c contiguous (rows
            // padded 16 -> 17 double2: conflict-free ds_write_b128 and ds_read_b128)
            // the slice is this wavefront's own: ordering within the wavefront only, never an s_barrier.
            // Wavefront-scope fences: the compiler may spread the writes over pass A's tail and start pass B
            // under the reads.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) ldsd[17 * (4 * rev16(s) + wp) + wc] = v[s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = ldsd[17 * t + c];

            // ---- pass B: radix-16 over c; slot s holds q' = rev16(s): bin k = 64 q' + t
            fft16_sel(v);

#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u == 15) {      // bin N-1 (lane 63) also feeds the DC slot, weight K - kf
                    const double pw = fma(v[u].y, v[u].y, v[u].x * v[u].x);
                    acc[u] = KONE ? pw : acc[u] + pw;
                    wdc = KONE ? pw : fma((double)(K - kf), pw, wdc);
                } else if constexpr (KONE) {
                    acc[u] = fma(v[u].y, v[u].y, v[u].x * v[u].x);
                } else {
                    acc[u] = fma(v[u].y, v[u].y, fma(v[u].x, v[u].x, acc[u]));
                }
            }
        }

        // ---- DC-slot rule (src/spectrum.c:25-33): slot N/2 (bin 0: lane 0, u = 0) takes
        // sum_k (K-k) * P_k[N-1] (bin N-1: lane 63, u = 15)
        {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, wdc);
            const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, 63);
            const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
            const double dcv = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            if (t == 0) acc[0] = dcv;
        }

        // ---- epilogue + store: slot u holds bin 64 q' + t, q' = rev16(u); fft-shift = flip the top
        // bit of the bin index = q' ^ 8.
        // Every store instruction writes 64 consecutive outputs, one per lane.  (Measured and not kept:
        // the row staged through the idle transposition buffer so that a lane stores 16 bytes -- 111.2 us
        // against 109.3 for these 4-byte-per-lane stores; profiles/r04_x1024_store_ab.txt.)
        {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long i = g * N + 64 * (rev16(u) ^ 8) + t;
                const double a = acc[u];
                if constexpr (OUT == OUT_PAYLOAD) {
                    // src/cbb_main.c:125-128, same operation order, in double
                    const double d = 10.0 * log10(fabs(p.lin_gain * a / (double)p.count));
                    const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
                    reinterpret_cast<uint8_t*>(p.out)[i] = (uint8_t)m;
                } else {
                    const double o = (OUT == OUT_DB) ? 10.0 * log10(a / (double)p.count) : a;
                    if constexpr (ROWF32) __builtin_nontemporal_store((float)o, reinterpret_cast<float*>(p.out) + i);
                    else __builtin_nontemporal_store(o, reinterpret_cast<double*>(p.out) + i);
                }
            }
        }
        g = g_next;
    }
}

// LDS: one 16 x 17 x 64-byte transposition slice per wavefront (+ the row counter)
constexpr size_t x_lds_bytes(int waves) { return (size_t)waves * 16 * 17 * 64 + (waves > 1 ? 16 : 0); }
size_t spectra_f64_1024x_lds_bytes(int waves) { return x_lds_bytes(waves >= 8 ? 8 : 1); }

template <int OUT, bool ROWF32, int WAVES>
static hipError_t launch_x_k(const SpectraParamsF64& p, int blocks, hipStream_t st)
{
    constexpr size_t lds_bytes = x_lds_bytes(WAVES);
    if constexpr (lds_bytes > 65536) {      // once per instantiation and device
        static std::atomic<bool> done[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!done[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectra_f64_1024x<OUT, true, ROWF32, WAVES>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e == hipSuccess && OUT == OUT_SUM)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectra_f64_1024x<OUT_SUM, false, ROWF32, WAVES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            done[dev].store(true, std::memory_order_release);
        }
    }
    if (blocks <= 0) return hipSuccess;      // rtlws_engine_prepare_f64: the attribute only, nothing enqueued
    if (p.k_avg == 1) {
        hipLaunchKernelGGL((spectra_f64_1024x<OUT, true, ROWF32, WAVES>), dim3(blocks), dim3(64 * WAVES), lds_bytes, st, p);
    } else if constexpr (OUT == OUT_SUM) {     // (dB / payload beside K-frame accumulators: spectrum_f64_fused.hip)
        hipLaunchKernelGGL((spectra_f64_1024x<OUT_SUM, false, ROWF32, WAVES>), dim3(blocks), dim3(64 * WAVES), lds_bytes, st, p);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int WAVES>
static hipError_t launch_x_w(const SpectraParamsF64& p, int blocks, hipStream_t st)
{
    switch (p.out_mode) {
    case OUT_SUM: return p.rows_f32 ? launch_x_k<OUT_SUM, true, WAVES>(p, blocks, st) : launch_x_k<OUT_SUM, false, WAVES>(p, blocks, st);
    case OUT_DB: return p.rows_f32 ? launch_x_k<OUT_DB, true, WAVES>(p, blocks, st) : launch_x_k<OUT_DB, false, WAVES>(p, blocks, st);
    default: return launch_x_k<OUT_PAYLOAD, false, WAVES>(p, blocks, st);
    }
}

// waves = 1: `blocks` one-wavefront workgroups; waves = 8: `blocks` workgroups of eight wavefronts (one per CU)
hipError_t launch_spectra_f64_1024x(const SpectraParamsF64& p, int blocks, int waves, hipStream_t st)
{
    return waves >= 8 ? launch_x_w<8>(p, blocks, st) : launch_x_w<1>(p, blocks, st);
}

}  // namespace rtlws
