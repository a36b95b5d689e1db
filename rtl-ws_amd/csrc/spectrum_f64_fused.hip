// spectrum_f64_fused.hip -- 1024- / 2048- / 4096-point cmplx_u8 frames -> power spectra in
// DOUBLE, at throughput: the batch form of what the reference computes per frame (src/spectrum.c:54-60
// convert to double, :21 f64 forward DFT, :23-34 |X|^2 + fft-shift + accumulate + DC-slot
// rule into a double buffer; K loop of src/cbb_main.c:50-59; dB / truncate / clamp of
// src/cbb_main.c:121-130 in double, same operation order).
//
// The f32 fused kernel holds "<= 1e-4 relative" only against a floor 50 dB under a row's
// maximum (f32 rounding next to a strong tone); the reference is f64 end to
// end.  This kernel is the same radix-16 x 16 x R3 structure as spectrum_fused.hip (R3 =
// N/256; N/16 threads per frame -- one 64-lane wavefront at N = 1024, where nothing needs a
// barrier, two at 2048, four at 4096 -- sixteen complex points per lane, two LDS
// transpositions) with every value a double: strict-metric error (floor 1e-9 of the row
// maximum) ~1e-12, i.e. rtlws_spectra_batch_f64 at batch rates instead of one
// workgroup-per-row radix-2 (spectrum_f64.hip, which keeps every other N and the generic CIC
// factors).  Round 4: compiled once per size (-DRTLWS_N=1024|2048|4096) for
//   * every input kind of spectrum.h -- cmplx_u8, cmplx_s32 (src/spectrum.c:72-76), real f32
//     (:90-94) -- and the CIC-fused input stage for R = 8 and the reference's own factors 10 and 12
//     (src/resample.c:21-40 feeding src/spectrum.c:65-81: BASELINE configs[3] in the reference's
//     precision); the input stages are frame_input.h, the integer block sums are exact in double;
//   * ROWF32: f64 arithmetic, rows rounded ONCE to f32 on the store (RTLWS_FLAG_ROWS_F32) -- the
//     contract's own byte count (2N + 4N/K per frame) with the strict-metric error
//     of one f32 rounding (<= 6e-8) instead of an f32 transform's.  A lane's four bins are then 16
//     bytes: one store instruction writes 1 KiB of consecutive bytes without any lane exchange.
//
// Cost model (MI355X): v_fma_f64 / v_add_f64 / v_mul_f64 issue at 4 cycles per wave64
// instruction -- half the f32 rate, and the rate ONE wavefront can issue at by itself, so 2
// wavefronts per SIMD are enough; ~420 of them (plus ~130 integer / address instructions) per
// 1024-point frame.  Algorithmic bytes: 2N in + 8N/K out = 10 240 B per 1024-point spectrum at
// K = 1; the kernel is bound by that traffic (4/5 of it stores), not by the
// arithmetic: 100 us per 65 536 frames with no stores, 135 us with them.
//
// LDS (double2 units, one buffer reused by both transpositions; tools/lds_sim.py f64;
// ROW = 17*R3 = 68 / 136 / 272):
//   transposition 1  (q1, m1) at q1*ROW + m1: rows of T padded to ROW, so that the pass-2
//       ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) land on
//       different 64-byte bank quarters;
//   transposition 2  (q1, m2, q2) at q2*ROW + (q1/J)*17 + (q1%J)*R3 + m2: a reader lane's
//       sixteen elements are contiguous, groups of 16 padded to 17.
// Every ds_write_b128 / ds_read_b128 of both is conflict-free at all three sizes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rtlws_internal.h"
#include "cic_lds.h"
#include "frame_input.h"
#include "fft_regs_f64.h"

#ifndef RTLWS_N
#error "compile with -DRTLWS_N=1024|2048|4096"
#endif

namespace rtlws {

typedef short pk_i16 __attribute__((ext_vector_type(2)));      // (re, im) of one integer point

using namespace f64;      // f2 = double2, real = double, fft16_fma, fft_last, hann_w ... in double

template <int N, int IN, bool WIN, int OUT, bool KONE, bool ROWF32>
__global__ __launch_bounds__(N / 16, 2) void spectra_f64_fused(const SpectraParamsF64 p)
{
    constexpr int T = N / 16, R3 = N / 256, J = 16 / R3;
    constexpr int F64F_ROW = 17 * R3;
    static_assert(!(ROWF32 && OUT == OUT_PAYLOAD), "payload rows are bytes in either form");
    static_assert(IN < IN_CU8_CICR_LDS4 || (N / 1024) * cic_stage_wave_bytes(IN) <= f64_fused_lds_bytes(N) - 16,
                  "the CIC staging slices live in the transposition buffer");
    extern __shared__ __attribute__((aligned(16))) double2 ldsd[];

    const int t = threadIdx.x;
    const int K = KONE ? 1 : p.k_avg;
    const long ngroups = p.ngroups;

    // first frame's bytes first (sixteen 2-byte loads per lane, one 128-byte line per instruction)
    unsigned raw[16];
    auto load_raw = [&](long frame) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(p.in) + frame * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[r] = __builtin_nontemporal_load(src + T * r + t);
    };
    if constexpr (IN == IN_CU8) {
        if ((long)blockIdx.x < ngroups) load_raw((long)blockIdx.x * K);
    }

    // the last pass's (cos, tan) pairs stay in registers, except in the one instantiation family whose
    // budget they break (4096-point, window, K-frame accumulators: 32 VGPRs of pairs): there the
    // workgroup copies the 2 KiB table into LDS once, behind the transposition buffer, and every frame
    // re-reads its eight pairs from there with the pass-3 LDS reads.  (Rounds 3-5 re-read them from the
    // L2-resident table in global memory through a laundered pointer, which compiled to FLAT loads: they
    // count on vmcnt AND lgkmcnt, so every wait for an LDS read also waited for an L2 round trip -- five
    // exposed round trips per frame, 230 us per 16 384 frames where 150 were due;
    // profiles/r06_f64_4096_tw3_from_lds.txt.)
    constexpr bool TW3_REGS = f64_fused_tw3_regs(N, IN, WIN, KONE);
    double2* const tw3_lds = ldsd + f64_fused_lds_elems(N);       // (!TW3_REGS) [16][R3/2] pairs
    if constexpr (!TW3_REGS) {
        for (int e = threadIdx.x; e < 16 * (R3 / 2); e += T) tw3_lds[e] = p.tw2f[e];
        __syncthreads();
    }
    // Hann on cmplx_u8 frames lives INSIDE the first butterfly layer of pass 1 (fft_regs_impl.h, hann_bfly4): the
    // layer's sums and differences are taken on the bytes (packed int16, exact), the weights enter as one FMA
    // per component where the unwindowed layer has an addition.  Its outputs are twice the windowed layer's:
    // the pass-1 twiddles carry the 1/2 (exact).
    constexpr bool HANN_BFLY = WIN && IN == IN_CU8;
    const double in_scale = HANN_BFLY ? 0.5 * p.in_scale : p.in_scale;
    f2 tw1[16], tw3[R3 / 2];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        tw1[s] = p.tw1f[t * 16 + s];
        if constexpr (HANN_BFLY) tw1[s] = mk(0.5 * tw1[s].x, 0.5 * tw1[s].y);
    }
#pragma unroll
    for (int m = 0; m < R3 / 2; ++m) tw3[m] = TW3_REGS ? p.tw2f[(t / R3) * (R3 / 2) + m] : mk(0.0, 0.0);
    // Hann weights from two lane constants (fft_regs_impl.h): sixteen registers pairs for K = 1 (four (cos, sin)
    // pairs for cmplx_u8 frames); with K-frame accumulators beside them the 256-VGPR budget is 2 short, so that
    // form regenerates them every frame (two FMAs each)
    constexpr bool WINREGS = WIN && KONE && N == 1024;
    f2 wcs = WIN ? p.hann_csf[t] : mk(0.0, 0.0);
    double win[16];
    f2 hcs[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) win[r] = (WINREGS && !HANN_BFLY) ? hann_w(r, wcs) : 1.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) hcs[m] = (WINREGS && HANN_BFLY) ? hann_cs_m(m, wcs) : mk(0.0, 0.0);
#pragma unroll
    for (int s = 1; s < 16; ++s) asm volatile("" ::"v"(tw1[s].x), "v"(tw1[s].y));
    if constexpr (WINREGS && !HANN_BFLY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(win[r]));
    }
    if constexpr (WINREGS && HANN_BFLY) {
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(hcs[m].x), "v"(hcs[m].y));
    }
    if constexpr (TW3_REGS) {
#pragma unroll
        for (int m = 0; m < R3 / 2; ++m) asm volatile("" ::"v"(tw3[m].x), "v"(tw3[m].y));
    }
    const int q1 = t / R3, m2 = t % R3;     // pass 2: (q1, m2); pass 3: (q2, g3) -- the same split

    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        double acc[16];
        double wdc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;

        for (int kf = 0; kf < K; ++kf) {
            const long frame = g * K + kf;
            f2 v[16];
            if constexpr (WIN && !WINREGS) asm volatile("" : "+v"(wcs.x), "+v"(wcs.y));   // not hoisted
            if constexpr (HANN_BFLY) {
                // the first butterfly layer of pass 1, on the bytes: (re, im) -> int16 pair by one v_perm_b32,
                // sums and differences packed (|.| <= 510; the 128 offset cancels in the differences and is
                // taken off the sums as 256), sign-extended conversions, then the weights (hann_bfly4)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    pk_i16 x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        x[j] = __builtin_bit_cast(pk_i16, __builtin_amdgcn_perm(raw[4 * j + m], raw[4 * j + m], 0x0c010c00u));
                    const pk_i16 off = {256, 256};
                    const pk_i16 e0 = x[0] + x[2] - off, e1 = x[0] - x[2], o0 = x[1] + x[3] - off, o1 = x[1] - x[3];
                    hann_bfly4(mk((double)e0.x, (double)e0.y), mk((double)e1.x, (double)e1.y),
                               mk((double)o0.x, (double)o0.y), mk((double)o1.x, (double)o1.y),
                               WINREGS ? hcs[m] : hann_cs_m(m, wcs), v[m], v[4 + m], v[8 + m], v[12 + m]);
                }
                long nf = frame + 1;
                if (kf + 1 == K) nf = (g + gridDim.x) * K;
                if (nf >= ngroups * K) nf = frame;        // in bounds, result unused
                load_raw(nf);
            } else if constexpr (IN == IN_CU8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (double)u8 is exact; the 128 offset is kept (it only reaches bin 0, which is never
                    // output: src/spectrum.c:31)
                    v[r] = mk((double)(raw[r] & 0xffu), (double)((raw[r] >> 8) & 0xffu));
                }
                long nf = frame + 1;
                if (kf + 1 == K) nf = (g + gridDim.x) * K;
                if (nf >= ngroups * K) nf = frame;        // in bounds, result unused
                load_raw(nf);
            } else {
                // cmplx_s32 / real f32 / CIC-fused block sums (frame_input.h): integers and floats,
                // exact in double; the 1/128 scale rides on the pass-1 twiddles as for cmplx_u8
                load_frame_points<N, IN>(p, frame, t, v, ldsd);
                if constexpr (WIN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const double w = WINREGS ? win[r] : hann_w(r, wcs);
                        v[r] = mk(v[r].x * w, v[r].y * w);
                    }
                }
            }

            // ---- pass 1: radix-16 over the slow digit, twiddle W_N^(m1*q1) (carries the 1/128)
            if constexpr (HANN_BFLY) fft16_fma_rows(v);
            else fft16_sel(v);
            v[0] = mk(v[0].x * in_scale, v[0].y * in_scale);
#pragma unroll
            for (int s = 1; s < 16; ++s) v[s] = cmul(v[s], tw1[s]);

            __syncthreads();   // one wavefront per workgroup: no s_barrier, only the LDS ordering
#pragma unroll
            for (int s = 0; s < 16; ++s) ldsd[rev16(s) * F64F_ROW + t] = v[s];
            __syncthreads();

            // ---- pass 2: lane (q1, m2) holds y[q1][4*r2 + m2]
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) v[r2] = ldsd[q1 * F64F_ROW + R3 * r2 + m2];
            fft16_sel(v);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; ++s)
                ldsd[rev16(s) * F64F_ROW + (q1 / J) * 17 + (q1 % J) * R3 + m2] = v[s];
            __syncthreads();

            // ---- pass 3: lane (q2, g3) = (t / R3, t % R3): sixteen contiguous elements, J
            // twiddled radix-R3 butterflies in fused-multiply-add form
            if constexpr (!TW3_REGS) {
                int o = (t / R3) * (R3 / 2);
                asm volatile("" : "+v"(o));           // not hoisted out of the frame loop (the registers are the point)
#pragma unroll
                for (int m = 0; m < R3 / 2; ++m) tw3[m] = tw3_lds[o + m];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = ldsd[q1 * F64F_ROW + m2 * 17 + i];
#pragma unroll
            for (int j = 0; j < J; ++j) fft_last<R3>(v, j * R3, tw3);

            // ---- |X|^2, accumulate; slot u = j*R3 + s holds bin k = 256*rev_last(s) + J*t + j
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u == 15) {      // bin N-1 (thread T-1) also feeds the DC slot, weight K - kf
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

        // ---- DC-slot rule (src/spectrum.c:25-33): slot N/2 (bin 0: thread 0, u = 0) takes
        // sum_k (K-k) * P_k[N-1] (bin N-1: thread T-1, u = 15)
        if constexpr (T == 64) {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, wdc);
            const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, 63);
            const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
            const double dcv = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            if (t == 0) acc[0] = dcv;
        } else {
            double* slot = reinterpret_cast<double*>(ldsd + f64_fused_lds_elems(N) - 1);
            __syncthreads();
            if (t == T - 1) *slot = wdc;
            __syncthreads();
            const double dcv = *slot;
            if (t == 0) acc[0] = dcv;
        }

        // ---- epilogue + store.  For each s the wavefront covers 256 consecutive outputs and a
        // lane owns four of them (bins 4t .. 4t+3): 32 bytes, i.e. two 16-byte stores that would
        // each leave every other 16 bytes of a line unwritten -- measured 1.58x the output bytes
        // at the HBM and 286 us per launch against 100 us with no stores at all.  The two halves
        // of the wavefront therefore trade pieces first (below), so that every store instruction
        // writes 1 KiB of consecutive bytes.  fft-shift = flip the top bit of the bin index.
        if constexpr (OUT == OUT_PAYLOAD) {
#pragma unroll
            for (int s = 0; s < R3; ++s) {
                const int i0 = 256 * (rev_last<R3>(s) ^ (R3 / 2)) + J * t;
                unsigned packed = 0;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    // src/cbb_main.c:125-128, same operation order, in double
                    const double d = 10.0 * log10(fabs(p.lin_gain * acc[j * R3 + s] / (double)p.count));
                    const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
                    packed |= m << (8 * j);
                }
                uint8_t* dst = reinterpret_cast<uint8_t*>(p.out) + g * N + i0;
                if constexpr (J == 4) *reinterpret_cast<unsigned*>(dst) = packed;
                else if constexpr (J == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)packed;
                else *dst = (uint8_t)packed;
            }
        } else if constexpr (ROWF32) {
            // f32 rows: the lane's J consecutive bins are 4*J bytes -- 16 / 8 / 4 per lane, every
            // store instruction consecutive bytes (1 KiB at N = 1024) with no lane exchange; one
            // rounding per value, after the dB conversion where there is one
#pragma unroll
            for (int s = 0; s < R3; ++s) {
                const int i0 = 256 * (rev_last<R3>(s) ^ (R3 / 2)) + J * t;
                float o[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    double a = acc[j * R3 + s];
                    if constexpr (OUT == OUT_DB) a = 10.0 * log10(a / (double)p.count);
                    o[j] = (float)a;
                }
                float* dst = reinterpret_cast<float*>(p.out) + g * N + i0;
                if constexpr (J == 4) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    const nt_f4 ov = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(ov, reinterpret_cast<nt_f4*>(dst));
                } else if constexpr (J == 2) {
                    typedef float nt_f2 __attribute__((ext_vector_type(2)));
                    const nt_f2 ov = {o[0], o[1]};
                    __builtin_nontemporal_store(ov, reinterpret_cast<nt_f2*>(dst));
                } else {
                    __builtin_nontemporal_store(o[0], dst);
                }
            }
        } else if constexpr (J == 4) {
            // N = 1024: a lane owns four consecutive bins = 32 bytes = 16-byte pieces A_t | B_t.
            // v_permlane32_swap exchanges the upper 32 lanes of A with the lower 32 lanes of B:
            // afterwards one register holds A_0..31 | B_0..31 -- the first 1 KiB of the block, every
            // 16-byte slot once -- and the other A_32..63 | B_32..63, the second KiB (lane l holds
            // slot 2*(l & 31) + (l >> 5) of its KiB).  No LDS, no wait.
            typedef double nt_d2 __attribute__((ext_vector_type(2)));
            const int slot = 2 * (t & 31) + (t >> 5);
#pragma unroll
            for (int s = 0; s < R3; ++s) {
                double o[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    o[j] = acc[j * R3 + s];
                    if constexpr (OUT == OUT_DB) o[j] = 10.0 * log10(o[j] / (double)p.count);
                }
                double x[2], y[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned long long ab = __builtin_bit_cast(unsigned long long, o[h]);
                    const unsigned long long bb = __builtin_bit_cast(unsigned long long, o[2 + h]);
                    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
                    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
                    x[h] = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
                    y[h] = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
                }
                nt_d2* dst = reinterpret_cast<nt_d2*>(reinterpret_cast<double*>(p.out) + g * N + 256 * (s ^ (R3 / 2)));
                const nt_d2 vx = {x[0], x[1]}, vy = {y[0], y[1]};
                __builtin_nontemporal_store(vx, dst + slot);
                __builtin_nontemporal_store(vy, dst + 64 + slot);
            }
        } else {
            // N = 2048 / 4096: a lane owns 2 / 1 consecutive bins per s: every store instruction
            // already writes consecutive bytes (16 / 8 per lane)
#pragma unroll
            for (int s = 0; s < R3; ++s) {
                const int i0 = 256 * (rev_last<R3>(s) ^ (R3 / 2)) + J * t;
                double o[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    o[j] = acc[j * R3 + s];
                    if constexpr (OUT == OUT_DB) o[j] = 10.0 * log10(o[j] / (double)p.count);
                }
                double* dst = reinterpret_cast<double*>(p.out) + g * N + i0;
                if constexpr (J == 2) {
                    typedef double nt_d2 __attribute__((ext_vector_type(2)));
                    const nt_d2 v2 = {o[0], o[1]};
                    __builtin_nontemporal_store(v2, reinterpret_cast<nt_d2*>(dst));
                } else {
                    __builtin_nontemporal_store(o[0], dst);
                }
            }
        }
    }
}

// ---- launch table (this file is compiled once per RTLWS_N) -----------------

// N = 4096 needs 69.6 KiB of LDS per workgroup: the attribute is set once per instantiation and
// device (a bit per device; the launch path itself makes no other HIP call than the launch)
template <int N, int IN, bool WIN, int OUT, bool KONE, bool ROWF32>
static hipError_t launch_f64f_one(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    constexpr size_t lds_bytes = f64_fused_lds_bytes(N) + (f64_fused_tw3_regs(N, IN, WIN, KONE) ? 0 : 16 * 16 * (N / 512));
    if constexpr (lds_bytes > 64 * 1024) {
        static std::atomic<unsigned long long> ready{0};
        const unsigned long long bit = 1ull << (device & 63);
        if (!(ready.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&spectra_f64_fused<N, IN, WIN, OUT, KONE, ROWF32>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            ready.fetch_or(bit, std::memory_order_release);
        }
    }
    if (blocks <= 0) return hipSuccess;      // rtlws_engine_prepare_f64: the attribute only, nothing enqueued
    hipLaunchKernelGGL((spectra_f64_fused<N, IN, WIN, OUT, KONE, ROWF32>), dim3(blocks), dim3(N / 16), lds_bytes, st, p);
    return hipGetLastError();
}

template <int N, int IN, bool WIN, int OUT, bool ROWF32>
static hipError_t launch_f64f_k(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    // K == 1 gets its own instantiation (no accumulators) on the kinds that have one in the f32
    // kernel too; cmplx_s32 / real f32 share the general-K code
    if constexpr (fused_kone_kind(IN)) {
        if (p.k_avg == 1) return launch_f64f_one<N, IN, WIN, OUT, true, ROWF32>(p, blocks, st, device);
    }
    return launch_f64f_one<N, IN, WIN, OUT, false, ROWF32>(p, blocks, st, device);
}

template <int N, int IN, bool WIN>
static hipError_t launch_f64f_o(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    switch (p.out_mode) {
    case OUT_SUM:
        return p.rows_f32 ? launch_f64f_k<N, IN, WIN, OUT_SUM, true>(p, blocks, st, device)
                          : launch_f64f_k<N, IN, WIN, OUT_SUM, false>(p, blocks, st, device);
    case OUT_DB:
        return p.rows_f32 ? launch_f64f_k<N, IN, WIN, OUT_DB, true>(p, blocks, st, device)
                          : launch_f64f_k<N, IN, WIN, OUT_DB, false>(p, blocks, st, device);
    default: return launch_f64f_k<N, IN, WIN, OUT_PAYLOAD, false>(p, blocks, st, device);
    }
}

template <int N, int IN>
static hipError_t launch_f64f_w(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    return p.window ? launch_f64f_o<N, IN, true>(p, blocks, st, device) : launch_f64f_o<N, IN, false>(p, blocks, st, device);
}

#define RTLWS_CAT2(a, b) a##b
#define RTLWS_CAT(a, b) RTLWS_CAT2(a, b)

hipError_t RTLWS_CAT(launch_spectra_f64_fused_, RTLWS_N)(const SpectraParamsF64& p, int in_kind, int blocks,
                                                         hipStream_t st, int device)
{
    constexpr int N = RTLWS_N;
    switch (in_kind) {
    case IN_CU8: return launch_f64f_w<N, IN_CU8>(p, blocks, st, device);
    case IN_CS32: return launch_f64f_w<N, IN_CS32>(p, blocks, st, device);
    case IN_RF32: return launch_f64f_w<N, IN_RF32>(p, blocks, st, device);
    case IN_CU8_CIC8: return launch_f64f_w<N, IN_CU8_CIC8>(p, blocks, st, device);
    case IN_CU8_CIC10: return launch_f64f_w<N, IN_CU8_CIC10>(p, blocks, st, device);
    case IN_CU8_CIC12: return launch_f64f_w<N, IN_CU8_CIC12>(p, blocks, st, device);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rtlws
