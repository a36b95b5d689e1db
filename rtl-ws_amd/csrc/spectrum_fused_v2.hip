// spectrum_fused_v2.hip -- the 4096- and 2048-point cmplx_u8 fused kernels with TWO
// virtual threads per lane: N/32 threads own one frame (two wavefronts at N = 4096, ONE at
// 2048 -- no barrier at all there) and every lane holds 32 complex points -- the same
// radix-16 x 16 x R3 decomposition, tables, twiddle forms and arithmetic as
// spectrum_fused.hip (whose "virtual thread" vt = 2t + h is one of that kernel's N/16
// threads), so results are bit-identical to it.
//
// Why: the N/16-thread form at these sizes joins two or four wavefronts at four
// __syncthreads() per frame, and a single wavefront can use at most every other issue slot
// of its SIMD.  Here
//   * a barrier joins half as many wavefronts (none at N = 2048), and there are half as
//     many LDS instructions: every access is 16 bytes (the two virtual threads of a lane
//     are adjacent in every index the transpositions use);
//   * each lane carries two independent radix-16 chains per pass;
//   * the 256-VGPR budget of 2 wavefronts per SIMD has room for the one-frame-ahead
//     prefetch of the raw bytes (sixteen 4-byte loads per lane).
// What that buys, measured: +5..8 % on K = 1 rows (rect_4096pt 0.53 ->
// 0.58, rect_2048pt 0.58 -> 0.61 of the HBM roofline) and nothing on K = 8 rows, which are
// bound by the energy of the transform at the package power cap, not by its schedule
// (§6.3) -- so the shim sends K = 1 descriptors here and keeps K > 1 on spectrum_fused.hip.
//
// Reference semantics: src/spectrum.c:47-63 (convert), :21 (forward DFT), :23-34
// (|X|^2, fft-shift, accumulate, DC-slot rule); the K loop of src/cbb_main.c:50-59;
// dB / payload epilogue of src/cbb_main.c:121-130.
//
// LDS layouts (float2 units; tools/lds_sim.py v2):
//   transposition 1  (q1, m1) at q1*(T+R3) + m1          -- as spectrum_fused.hip
//   transposition 2  (q1, m2, q2) at q2*A2 + (q1/J)*18 + (q1%J)*R3 + m2 -- groups of 16
//       padded to 18 as there, the q2 stride A2 = 290 (N = 4096) / 146 (N = 2048) instead
//       of 18*R3, chosen with the pass-3 assignment of pairs to lanes (pair_of_lane) so
//       that every ds_read_b128 lane group -- 16 lanes {0-3,12-15,20-27}, ... -- covers the
//       64 banks exactly once.  The permutation costs nothing: a wavefront still stores
//       512 (N = 4096) or 1 024 (N = 2048) contiguous bytes per instruction, in a
//       permuted lane order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtlws_internal.h"
#include "fft_regs.h"

namespace rtlws {

#define RTLWS_V2_PREFETCH 1      // raw bytes of the next frame in flight during the transform
#define RTLWS_V2_WINREGS 0       // 1: the 32 Hann weights of a lane live in registers (12 VGPRs spill with K > 1
                                 // and the prefetch); 0: regenerated per frame from 4 lane constants (236 VGPRs)
#define RTLWS_V2_NT_STORE 1      // nontemporal float2 stores: +2.5..3.5 % (rows are written once, never re-read)

constexpr int V2_P2 = 18;        // transposition 2: padded group of 16
constexpr int v2_s1(int n_fft) { return n_fft / 16 + n_fft / 256; }      // transposition 1 row stride (T + R3)

// pass 3: which pair of adjacent virtual threads (2*pair, 2*pair + 1) a lane owns
template <int N>
__device__ __forceinline__ int pair_of_lane(int t)
{
    const int l = t & 31;
    if constexpr (N == 4096) {
        // 128 pairs, q2 = pair / 8: each read group holds two sets of eight consecutive pairs
        // from q2 rows of different parity
        const int s = l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
        return (t & ~31) | s;
    } else {
        // 64 pairs, q2 = pair / 4: read group k (of the guide's four) holds pairs 8k .. 8k+7 and
        // 32+8k .. 32+8k+7, i.e. q2 rows {2k, 2k+1, 8+2k, 9+2k}
        const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28;
        const int idx = g1 ? (l < 12 ? l - 4 : l < 20 ? l - 8 : l - 16) : (l < 4 ? l : l < 16 ? l - 8 : l - 12);
        const int k = 2 * (t >> 5) + (g1 ? 1 : 0);
        return idx < 8 ? 8 * k + idx : 32 + 8 * k + (idx - 8);
    }
}

template <int N>
__device__ __forceinline__ void load_raw_v2(const SpectraParams& p, long frame, int t, unsigned (&raw)[16])
{
    constexpr int T = N / 16;
    // x[T*r + 2t + h], h = 0, 1: one 4-byte load; a wave-instruction covers 256 contiguous bytes
    const unsigned* src = reinterpret_cast<const unsigned*>(p.in) + frame * (N / 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = __builtin_nontemporal_load(src + (T / 2) * r + t);
}

template <bool WIN>
__device__ __forceinline__ void convert_v2(const unsigned (&raw)[16], const float (&win)[2][16],
                                           const float2 (&wcs)[2], f2 (&v)[2][16])
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float c[4];
        // one conversion instruction per component (see convert_u8 in spectrum_fused.hip)
        asm("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(c[0]) : "v"(raw[r]));
        asm("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(c[1]) : "v"(raw[r]));
        asm("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(c[2]) : "v"(raw[r]));
        asm("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(c[3]) : "v"(raw[r]));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (WIN) {
                const float w = RTLWS_V2_WINREGS ? win[h][r] : hann_w(r, wcs[h]);
                const float o = -128.0f * w;                       // exact: (x - 128) * w as one FMA
                v[h][r] = mk(fmaf(c[2 * h], w, o), fmaf(c[2 * h + 1], w, o));
            } else {
                v[h][r] = mk(c[2 * h], c[2 * h + 1]);              // offset kept: only bin 0 sees it
            }
        }
    }
}

__device__ __forceinline__ void st_pair(float2* dst, f2 a, f2 b)
{
    *reinterpret_cast<float4*>(dst) = make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void ld_pair(const float2* src, f2& a, f2& b)
{
    const float4 x = *reinterpret_cast<const float4*>(src);
    a = mk(x.x, x.y);
    b = mk(x.z, x.w);
}

template <int N, bool WIN, int OUT, bool KONE>
__global__ __launch_bounds__(N / 32, 2) void spectra_fused_v2(const SpectraParams p)
{
    static_assert(N == 4096 || N == 2048, "pass-3 lane maps and strides exist for these two sizes");
    constexpr int T = N / 16;      // virtual threads per frame
    constexpr int R3 = N / 256;    // radix of the last pass
    constexpr int J = 16 / R3;     // last-pass butterflies per virtual thread
    constexpr int V2_S1 = v2_s1(N), V2_A2 = v2_a2(N);
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int t = threadIdx.x;     // owns virtual threads 2t, 2t + 1 in passes 1 and 2
    const int K = KONE ? 1 : p.k_avg;
    const long ngroups = p.ngroups;

    unsigned raw[16];
    if constexpr (RTLWS_V2_PREFETCH) {
        if ((long)blockIdx.x < ngroups) load_raw_v2<N>(p, (long)blockIdx.x * K, t, raw);
    }

    // lane constants, resident for every frame of this (persistent) workgroup
    f2 tw1[2][16], tw3[R3 / 2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 16; ++s) tw1[h][s] = p.tw1[(2 * t + h) * 16 + s];
    const int pr = pair_of_lane<N>(t);                // pass 3: virtual threads 2*pr, 2*pr + 1
#pragma unroll
    for (int m = 0; m < R3 / 2; ++m) tw3[m] = p.tw2[((2 * pr) / R3) * (R3 / 2) + m];     // q2 = (2*pr) / R3
    float2 wcs[2];
    float win[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wcs[h] = WIN ? p.hann_cs[2 * t + h] : make_float2(0.0f, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) win[h][r] = (WIN && RTLWS_V2_WINREGS) ? hann_w(r, wcs[h]) : 1.0f;
    }
    // retire the table loads before the loop (see spectrum_fused.hip)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int s = 1; s < 16; ++s) asm volatile("" ::"v"(tw1[h][s].x), "v"(tw1[h][s].y));
        if constexpr (WIN && RTLWS_V2_WINREGS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(win[h][r]));
        }
    }
#pragma unroll
    for (int m = 0; m < R3 / 2; ++m) asm volatile("" ::"v"(tw3[m].x), "v"(tw3[m].y));
    const float in_scale = p.in_scale;

    const int q1 = (2 * t) / R3, mb = (2 * t) % R3;   // passes 1 -> 2: (q1, m2 = mb + h)
    const int q2 = (2 * pr) / R3, gb = (2 * pr) % R3; // pass 3: (q2, g3 = gb + h)

    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        float acc[2][16];
        float wdc = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[h][u] = 0.0f;

        for (int kf = 0; kf < K; ++kf) {
            const long frame = g * K + kf;
            f2 v[2][16];
            if constexpr (WIN && !RTLWS_V2_WINREGS) {
                // the weights are regenerated from the lane constants every frame (two FMAs each):
                // opaque here, or hipcc hoists all thirty-two out of the loop and spills
                asm volatile("" : "+v"(wcs[0].x), "+v"(wcs[0].y), "+v"(wcs[1].x), "+v"(wcs[1].y));
            }
            if constexpr (RTLWS_V2_PREFETCH) {
                convert_v2<WIN>(raw, win, wcs, v);
                // all thirty-two conversions first: the next frame's loads reuse raw[] (left to
                // itself hipcc starts them early and keeps both sets of bytes live)
                __builtin_amdgcn_sched_barrier(0);
                long nf = frame + 1;
                if (kf + 1 == K) nf = (g + gridDim.x) * K;
                if (nf >= ngroups * K) nf = frame;        // in bounds, result unused
                load_raw_v2<N>(p, nf, t, raw);
            } else {
                load_raw_v2<N>(p, frame, t, raw);
                convert_v2<WIN>(raw, win, wcs, v);
            }

            // ---- pass 1: radix-16 over the slow digit, twiddle W_N^(m1*q1), m1 = 2t + h
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                fft16_sel(v[h]);
                v[h][0] = mk(v[h][0].x * in_scale, v[h][0].y * in_scale);
#pragma unroll
                for (int s = 1; s < 16; ++s) v[h][s] = cmul(v[h][s], tw1[h][s]);
            }
            __syncthreads();   // previous frame's pass-3 reads are done
#pragma unroll
            for (int s = 0; s < 16; ++s) st_pair(lds + rev16(s) * V2_S1 + 2 * t, v[0][s], v[1][s]);
            __syncthreads();

            // ---- pass 2: virtual thread (q1, m2): y[q1][16*r2 + m2]
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) ld_pair(lds + q1 * V2_S1 + R3 * r2 + mb, v[0][r2], v[1][r2]);
            fft16_sel(v[0]);
            fft16_sel(v[1]);
            __syncthreads();   // everyone has read transposition 1
#pragma unroll
            for (int s = 0; s < 16; ++s)
                st_pair(lds + rev16(s) * V2_A2 + (q1 / J) * V2_P2 + (q1 % J) * R3 + mb, v[0][s], v[1][s]);
            __syncthreads();

            // ---- pass 3: virtual thread (q2, g3): sixteen contiguous elements, twiddled radix-16
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    ld_pair(lds + q2 * V2_A2 + (gb + h) * V2_P2 + 2 * i, v[h][2 * i], v[h][2 * i + 1]);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                fft_last<R3>(v[0], j * R3, tw3);
                fft_last<R3>(v[1], j * R3, tw3);
            }

            // ---- |X|^2, accumulate; slot u = j*R3 + s holds bin k = 256*rev_last(s) + J*(2*pr + h) + j
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (u == 15) {   // bin N-1 (pair T/2-1, h = 1) also feeds the DC slot, weight K - kf
                        // (slot 15 of EVERY virtual thread in this form: bit-identical to spectrum_fused.hip)
                        const float pw = fmaf(v[h][u].y, v[h][u].y, v[h][u].x * v[h][u].x);
                        acc[h][u] = KONE ? pw : acc[h][u] + pw;
                        if (h == 1) wdc = KONE ? pw : fmaf((float)(K - kf), pw, wdc);
                    } else if constexpr (KONE) {
                        acc[h][u] = fmaf(v[h][u].y, v[h][u].y, v[h][u].x * v[h][u].x);
                    } else {
                        acc[h][u] = fmaf(v[h][u].y, v[h][u].y, fmaf(v[h][u].x, v[h][u].x, acc[h][u]));
                    }
                }
        }

        // ---- DC-slot rule (reference src/spectrum.c:25-33): slot N/2 (bin 0: pair 0, h = 0,
        // u = 0) takes sum_k (K-k) * P_k[N-1] (bin N-1: pair T/2-1, h = 1, u = 15)
        {
            float* slot = reinterpret_cast<float*>(lds + v2_lds_f2(N) - 2);
            __syncthreads();
            if (pr == T / 2 - 1) *slot = wdc;
            __syncthreads();
            const float dcv = *slot;
            if (pr == 0) acc[0][0] = dcv;
        }

        // ---- epilogue + store: for each s the workgroup covers 256 consecutive outputs, a lane
        // 2*J of them (bins J*(2*pr + h) + j); fft-shift = flip the top bit of the bin index
#pragma unroll
        for (int s = 0; s < R3; ++s) {
            const int i0 = 256 * (rev_last<R3>(s) ^ (R3 / 2)) + 2 * J * pr;
            float o[2 * J];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float a = acc[h][j * R3 + s];
                    constexpr float DB_PER_LOG2 = 3.01029995663981195f;
                    if constexpr (OUT == OUT_DB) a = fmaf(DB_PER_LOG2, __builtin_amdgcn_logf(a), p.db_offset);
                    if constexpr (OUT == OUT_PAYLOAD) a = DB_PER_LOG2 * __builtin_amdgcn_logf(fabsf(a * p.lin_gain));
                    o[h * J + j] = a;
                }
            if constexpr (OUT == OUT_PAYLOAD) {
                unsigned packed = 0;
#pragma unroll
                for (int i = 0; i < 2 * J; ++i) {
                    const float d = o[i];      // (int) truncation toward zero, then clamp; NaN/-inf -> 0
                    const unsigned m = (d >= 0.0f) ? (d <= 255.0f ? (unsigned)(int)d : 255u) : 0u;
                    packed |= m << (8 * i);
                }
                uint8_t* dst = reinterpret_cast<uint8_t*>(p.out) + g * N + i0;
                if constexpr (J == 2) *reinterpret_cast<unsigned*>(dst) = packed;
                else *reinterpret_cast<uint16_t*>(dst) = (uint16_t)packed;
            } else {
                float* dst = reinterpret_cast<float*>(p.out) + g * N + i0;
                if constexpr (J == 2) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    const nt_f4 ov = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(ov, reinterpret_cast<nt_f4*>(dst));
                } else {
                    typedef float nt_f2 __attribute__((ext_vector_type(2)));
                    const nt_f2 ov = {o[0], o[1]};
                    __builtin_nontemporal_store(ov, reinterpret_cast<nt_f2*>(dst));
                }
            }
        }
    }
}

template <int N, bool WIN, int OUT>
static hipError_t launch_v2_k(const SpectraParams& p, int blocks, hipStream_t st)
{
    const size_t lds_bytes = v2_lds_bytes(N);
    if (p.k_avg == 1)
        hipLaunchKernelGGL((spectra_fused_v2<N, WIN, OUT, true>), dim3(blocks), dim3(N / 32), lds_bytes, st, p);
    else
        hipLaunchKernelGGL((spectra_fused_v2<N, WIN, OUT, false>), dim3(blocks), dim3(N / 32), lds_bytes, st, p);
    return hipGetLastError();
}

template <int N, bool WIN>
static hipError_t launch_v2_o(const SpectraParams& p, int blocks, hipStream_t st)
{
    switch (p.out_mode) {
    case OUT_SUM: return launch_v2_k<N, WIN, OUT_SUM>(p, blocks, st);
    case OUT_DB: return launch_v2_k<N, WIN, OUT_DB>(p, blocks, st);
    default: return launch_v2_k<N, WIN, OUT_PAYLOAD>(p, blocks, st);
    }
}

hipError_t launch_spectra_fused_v2(const SpectraParams& p, int blocks, hipStream_t st)
{
    if (p.n_fft == 2048)
        return p.window ? launch_v2_o<2048, true>(p, blocks, st) : launch_v2_o<2048, false>(p, blocks, st);
    return p.window ? launch_v2_o<4096, true>(p, blocks, st) : launch_v2_o<4096, false>(p, blocks, st);
}

}  // namespace rtlws
