// rtlws_internal.h -- shared between the HIP translation units of librtlws_hip.
#ifndef RTLWS_INTERNAL_H
#define RTLWS_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtlws {

// Input kinds as the kernels see them (the public enum rtlws_input plus the
// CIC-fused variants selected from rtlws_spectra_desc::cic_r).
// (IN_CU8_CICRw: generic R read w bytes at a time; keep them last and in this order)
// IN_CU8_CICR_LDSg: input staged through LDS by global_load_lds (whole
// 128-byte lines whatever R is), g of a wavefront's sixteen 64-sample pieces at
// a time.
enum { IN_CU8 = 0, IN_CS32 = 1, IN_RF32 = 2, IN_CU8_CIC8 = 3,
       IN_CU8_CICR2 = 4, IN_CU8_CICR4 = 5, IN_CU8_CICR8 = 6, IN_CU8_CICR16 = 7,
       IN_CU8_CICR_LDS4 = 8, IN_CU8_CICR_LDS2 = 9, IN_CU8_CICR_LDS1 = 10,
       // the reference's own factors (sample_rate / 192000, src/main.c:23,154), staged
       // through LDS like LDS4 but with the factor a compile-time constant
       IN_CU8_CIC10 = 11, IN_CU8_CIC12 = 12 };
constexpr int cic_ct_factor(int in_kind) { return in_kind == IN_CU8_CIC10 ? 10 : in_kind == IN_CU8_CIC12 ? 12 : 0; }

// LDS staging of the CIC-fused input: each wavefront owns CICR_LDS_WAVE_BYTES of
// the transposition buffer (fused_lds_f2 gives >= 9216 B per wavefront at every
// N); a piece (64 samples) is 128R bytes.
constexpr int CICR_LDS_WAVE_BYTES = 9216;
// pieces in flight per round for the compile-time factors (8: measured +3 % over 4 at R = 12, N = 2048, although the 24 KiB of LDS per workgroup leave 3 wavefronts per SIMD instead of 4: bytes in flight matter more than wavefronts; -DRTLWS_CIC_CT_ROUND=4 for the A/B)
#define RTLWS_CIC_CT_ROUND 8
constexpr int cicr_lds_round(int in_kind) { return in_kind >= IN_CU8_CIC10 ? RTLWS_CIC_CT_ROUND : in_kind == IN_CU8_CICR_LDS4 ? 4 : in_kind == IN_CU8_CICR_LDS2 ? 2 : 1; }
// bytes of LDS one wavefront stages its round in
constexpr int cic_stage_wave_bytes(int in_kind)
{
    return cic_ct_factor(in_kind) ? cicr_lds_round(in_kind) * 128 * cic_ct_factor(in_kind) : CICR_LDS_WAVE_BYTES;
}
constexpr int cicr_lds_max_r(int round) { return CICR_LDS_WAVE_BYTES / (round * 128); }   // 18, 36, 72

// generic-R CIC input kind: per-lane direct loads by the alignment of a 2R-byte
// decimated sample, or LDS staging (3 <= R <= 72; measured 1.2-5x faster than
// the direct loads there, slower at R = 2: tools/cic_fused_rates.py)
constexpr int cicr_direct_kind(int R) { return (R % 8 == 0) ? IN_CU8_CICR16 : (R % 4 == 0) ? IN_CU8_CICR8 : (R % 2 == 0) ? IN_CU8_CICR4 : IN_CU8_CICR2; }
constexpr int cicr_lds_kind(int R, int round)   // -1: does not fit
{
    return (R < 2 || R > cicr_lds_max_r(round)) ? -1 : round == 4 ? IN_CU8_CICR_LDS4 : round == 2 ? IN_CU8_CICR_LDS2 : IN_CU8_CICR_LDS1;
}
constexpr int cicr_kind(int R)
{
    if (R == 10) return IN_CU8_CIC10;
    if (R == 12) return IN_CU8_CIC12;
    if (R < 3 || R > cicr_lds_max_r(1)) return cicr_direct_kind(R);
    return R <= cicr_lds_max_r(4) ? IN_CU8_CICR_LDS4 : R <= cicr_lds_max_r(2) ? IN_CU8_CICR_LDS2 : IN_CU8_CICR_LDS1;
}
enum { OUT_SUM = 0, OUT_DB = 1, OUT_PAYLOAD = 2 };

struct SpectraParams {
    const void* in;        // device: frames
    void* out;             // device: rows
    long ngroups;          // output rows (= nframes / k_avg)
    int k_avg;
    int cic_r;             // 1 when unused
    int n_fft;
    int out_mode;          // OUT_*
    const float2* tw1;     // fused: [T][16] scale * W_N^(m1*rev16(s)); direct: [N] W_N^e
    const float2* tw2;     // fused: [16][R3/2] last-pass (cos, sin/cos) pairs of W_T^q2
    const float* window;   // [N] or nullptr (direct kernel reads it; fused: only a flag)
    const float2* hann_cs; // fused: [T] (0.5*cos, 0.5*sin)(2*pi*t/N), the lane constants of the generated Hann
    float db_offset;       // -10*log10(K)
    float lin_gain;        // gain / K for the payload epilogue
    float in_scale;        // 1/128 or 1 (fused: also folded into tw1[s >= 1])
};

// The f64 ("exact") kernel of spectrum_f64.hip: reference-precision arithmetic
// for the reference-API paths (spectrum.h, cbb_main.h).
struct SpectraParamsF64 {
    const void* in;        // device: frames
    void* out;             // device: rows (f64, or u8 for OUT_PAYLOAD)
    long ngroups;
    int k_avg;
    int cic_r;             // 1 when unused
    int n_fft;             // 2 .. 8192
    int log2n;             // log2(n_fft) when it is a power of two, else 0 (direct sum)
    int out_mode;          // OUT_*
    int count;             // divisor of the dB / payload epilogues (= k_avg)
    const double2* tw;     // [N] W_N^k, built in long double, rounded once
    const double* window;  // [N] or nullptr
    double lin_gain;       // 10^(gain_db/10), C integer division (src/cbb_main.c:112)
    double in_scale;       // 1/128 or 1
    // spectrum_f64_fused.hip (1024-point cmplx_u8 frames): the fused kernel's tables in double
    const double2* tw1f;     // [64][16] scale * W_N^(t*rev16(s))
    const double2* tw2f;     // [16][2] last-pass (cos, sin/cos) pairs
    const double2* hann_csf; // [64] (0.5*cos, 0.5*sin)(2*pi*t/N)
    int rows_f32;            // RTLWS_FLAG_ROWS_F32: f64 arithmetic, rows rounded once to f32 on the store
    // spectrum_f64_1024x.hip (1024 = 4 x 16 x 16, one LDS transposition)
    const double2* twxa;     // [4][8] (cos, tan) pairs of pass A's geometric pre-twiddle alpha = W_64^p
    const double2* twxb;     // [64][16] W_1024^(c (4 q + p)) / 128, lane 16 p + c, slot s: q = rev16(s)
};

// which f64 descriptors take the fused throughput kernel (the rest: spectrum_f64.hip): the three
// fused sizes with any input kind of spectrum.h, or cmplx_u8 through the CIC-fused input stage
// for R = 8 and the reference's own factors 10 and 12 (in_kind: the kernels' IN_* value)
constexpr bool f64_fused_in_kind(int in_kind)
{
    return in_kind == IN_CU8 || in_kind == IN_CS32 || in_kind == IN_RF32 || in_kind == IN_CU8_CIC8 ||
           in_kind == IN_CU8_CIC10 || in_kind == IN_CU8_CIC12;
}
constexpr bool f64_fused_kind(int n_fft, int in_kind)
{
    return (n_fft == 1024 || n_fft == 2048 || n_fft == 4096) && f64_fused_in_kind(in_kind);
}
// the last pass's (cos, tan) pairs stay in registers, except in the instantiations whose budget
// (256 VGPRs at 2 wavefronts per SIMD) they break: those re-read them every frame, with the pass-3 LDS
// reads, from a 2 KiB copy of the table behind the transposition buffer
constexpr bool f64_fused_tw3_regs(int n_fft, int in_kind, bool win, bool kone)
{
    return !(n_fft == 4096 && win && (!kone || in_kind == IN_CU8));
}
// LDS in double2 elements: 16 rows of 17*R3 (the larger of the two transpositions ends at
// 15*17*R3 + (R3-1)*17 + 16) + one element for the DC slot of the multi-wavefront sizes
constexpr int f64_fused_lds_elems(int n_fft) { return 15 * 17 * (n_fft / 256) + (n_fft / 256 - 1) * 17 + 16 + 1; }
constexpr int f64_fused_lds_bytes(int n_fft) { return 16 * f64_fused_lds_elems(n_fft); }
// 8 wavefronts per CU (2 per SIMD) at every size: 8 / 4 / 2 workgroups
constexpr int f64_fused_blocks_per_cu(int n_fft) { return 8 / (n_fft / 1024); }
// rectangular 1024-point cmplx_u8 frames: the one-transposition kernel (engine option f64_x1024)
hipError_t launch_spectra_f64_1024x(const SpectraParamsF64&, int blocks, int waves, hipStream_t);
size_t spectra_f64_1024x_lds_bytes(int waves);      // dynamic LDS per workgroup of that form (waves = 1 | 8)
hipError_t launch_spectra_f64_fused_1024(const SpectraParamsF64&, int in_kind, int blocks, hipStream_t, int device);
hipError_t launch_spectra_f64_fused_2048(const SpectraParamsF64&, int in_kind, int blocks, hipStream_t, int device);
hipError_t launch_spectra_f64_fused_4096(const SpectraParamsF64&, int in_kind, int blocks, hipStream_t, int device);

// Occupancy the fused kernel is built for (waves per SIMD = __launch_bounds__'
// second argument), by instantiation, chosen so that NO instantiation spills
// (tests/test_abi_cpu.py reads the code-object metadata and fails on any
// vgpr_spill_count > 0).  4 waves/SIMD = 128 VGPRs, 3 = 168.
//   N = 1024: the rectangular K = 1 kernels (117-121 VGPRs) and every
//     rectangular kind without prefetch registers or accumulators fit 4; a
//     window (16 more registers) or K > 1 on the two prefetching kinds needs 3.
//     The headline kernel is LAUNCHED at 2 (below): an occupancy choice, not a
//     register limit.
//   N = 2048: the rectangular K = 1 u8 kernel and the CIC-fused rectangular kinds fit
//     4 and are faster there; the rest carry R3 = 8 last-pass twiddles plus a window
//     or accumulators and are built for 3.
//   N = 4096: 3 (R3 = 16 twiddles and a bigger last pass).
// RTLWS_WAVES_BIG overrides the "3" for experiments.
#define RTLWS_WAVES_BIG 3
// input kinds that get a dedicated K == 1 instantiation (no accumulators)
constexpr bool fused_kone_kind(int in_kind)
{
    return in_kind == IN_CU8 || in_kind == IN_CU8_CIC8 || in_kind >= IN_CU8_CIC10;
}
constexpr int fused_lds_f2(int n_fft);
constexpr int fused_lds_bytes(int n_fft, int in_kind, bool win = false);
constexpr int fused_waves_per_simd(int n_fft, int in_kind, bool win, bool kone)
{
    const bool acc_and_prefetch = !kone && fused_kone_kind(in_kind);
    // (two instantiation families want ~171 VGPRs and the allocator spills 18 of them
    // at the 168 cap instead of finding the 3 it is short of: built for 2)
    if (n_fft == 2048 && in_kind >= IN_CU8_CIC10 && win && !kone) return 2;
    // the rectangular K = 1 2048-point u8 kernel fits 128 VGPRs (122) and is 7 % faster
    // at 4 wavefronts per SIMD (0.59 against 0.55 of the HBM roofline, same call)
    if (n_fft == 2048 && in_kind == IN_CU8 && !win && kone) return 4;
    // (the rectangular cmplx_s32 / real-f32 input kinds of spectrum.h fit as well: 127 / 124)
    if (n_fft == 2048 && (in_kind == IN_CS32 || in_kind == IN_RF32) && !win) return 4;
    // The headline kernel (1024-point, u8, rectangular, K = 1; 117 VGPRs) runs 8 one-wave
    // workgroups per CU, 2 per SIMD, 32 rows each at 65 536 rows: measured 0.66-0.67
    // against 0.63-0.65 for 16 on one box, +1 % on two others, never slower, and steadier
    // from run to run (12, 10 and 7 per CU are in between, 9 -- uneven over the four
    // SIMDs -- and 6 or fewer are slower).  The kernel is at the package power cap
    // either way; fewer resident wavefronts queue less on the LDS pipe.
    if (n_fft == 1024 && in_kind == IN_CU8 && !win && kone) return 2;
    const int by_regs = (n_fft == 1024 && !win && !acc_and_prefetch) ? 4
                        : (n_fft == 2048 && !win && in_kind >= IN_CU8_CIC8 && !acc_and_prefetch) ? 4
                        : RTLWS_WAVES_BIG;
    // workgroups per CU that fit the 160 KiB LDS, n_fft/1024 wavefronts each, over 4 SIMDs
    const int by_lds = (163840 / fused_lds_bytes(n_fft, in_kind, win)) * (n_fft / 1024) / 4;
    return by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
}

// One-frame-ahead prefetch of the raw cmplx_u8 bytes (16 VGPRs).  Worth <= 2 % at
// 12-16 resident wavefronts per CU; the windowed 4096-point kernels have no
// room for it at 168 VGPRs, so they load in the loop instead of spilling.
#define RTLWS_PREFETCH_4096WIN 0
constexpr bool fused_prefetch_u8(int n_fft, bool win) { return RTLWS_PREFETCH_4096WIN || !(n_fft == 4096 && win); }

// ... those kernels prefetch WITHOUT registers instead: the next frame's bytes are copied
// into LDS by global_load_lds (LDS-DMA, two 1-KiB copies per wavefront and frame, double
// buffered: 4*N bytes of LDS beside the transposition buffer) while the current frame is
// transformed, and read back with sixteen ds_read_u16 per thread.  -DRTLWS_DMA_PF=0: off.
#define RTLWS_DMA_PF 1
constexpr bool fused_dma_prefetch(int n_fft, int in_kind, bool win)
{
    return RTLWS_DMA_PF && in_kind == IN_CU8 && !fused_prefetch_u8(n_fft, win);
}

// LDS the fused kernel needs, in float2 units: 16 (padded) rows + one spare slot
// (layouts: spectrum_fused.hip, "LDS layouts").
constexpr int fused_lds_f2(int n_fft)
{
    return 16 * 18 * (n_fft / 256) + 2;     // transposition 2 is the larger of the two
}
// ... in bytes, by input kind: the CIC staging slices share the buffer and may be the larger need
constexpr int fused_lds_bytes(int n_fft, int in_kind, bool win)
{
    const int tr = 8 * fused_lds_f2(n_fft);
    const int st = in_kind >= IN_CU8_CICR_LDS4 ? (n_fft / 1024) * cic_stage_wave_bytes(in_kind) : 0;
    const int pf = fused_dma_prefetch(n_fft, in_kind, win) ? 2 * 2 * n_fft : 0;     // two raw frames
    return (st > tr ? st : tr) + pf;
}

// ---- spectrum_fused_v2.hip: 4096- / 2048-point cmplx_u8 frames, two virtual threads per lane ----
// transposition 2: q2 stride (chosen with the pass-3 lane map so that reads are conflict-free)
constexpr int v2_a2(int n_fft) { return n_fft == 4096 ? 290 : 146; }
// LDS in float2 units: transposition 2 is the larger (q2 stride A2, groups of 16 padded to 18) + the DC slot
constexpr int v2_lds_f2(int n_fft)
{
    return 15 * v2_a2(n_fft) + (n_fft / 256 - 1) * 18 + (16 / (n_fft / 256) - 1) * (n_fft / 256) + (n_fft / 256) + 2;
}
constexpr int v2_lds_bytes(int n_fft) { return 8 * v2_lds_f2(n_fft); }
// 2 wavefronts per SIMD (256 VGPRs) = 8 per CU: 4 two-wavefront workgroups at N = 4096, 8 one-wavefront
// ones at 2048 -- which is also what the LDS holds
constexpr int v2_blocks_per_cu(int n_fft)
{
    return 163840 / v2_lds_bytes(n_fft) < 8 / (n_fft / 2048) ? 163840 / v2_lds_bytes(n_fft) : 8 / (n_fft / 2048);
}
// which descriptors take it (the shim may override with RTLWS_V2=0|1 for A/B runs)
#define RTLWS_V2_DEFAULT 1
constexpr bool fused_v2_kind(int n_fft, int in_kind) { return (n_fft == 4096 || n_fft == 2048) && in_kind == IN_CU8; }
hipError_t launch_spectra_fused_v2(const SpectraParams&, int blocks, hipStream_t);

hipError_t launch_spectra_fused_1024(const SpectraParams&, int in_kind, int blocks, hipStream_t);
hipError_t launch_spectra_fused_2048(const SpectraParams&, int in_kind, int blocks, hipStream_t);
hipError_t launch_spectra_fused_4096(const SpectraParams&, int in_kind, int blocks, hipStream_t);
hipError_t launch_spectra_direct(const SpectraParams&, int in_kind, hipStream_t);

hipError_t launch_cic_block_sums(int R, const void* d_src, long dst_len, void* d_dst, hipStream_t, int cus);
hipError_t launch_halfband(const float* d_x, float* d_y, long out_len, hipStream_t, int cus);
hipError_t launch_fm_demod(const void* d_iq, long len, const float* d_prev_in, float* d_prev_out,
                           float* d_out, hipStream_t, int cus);
hipError_t launch_clock_probe(const int* stop_flag, unsigned long long* out, int max_polls, hipStream_t);
hipError_t launch_clock_stamp(unsigned long long* d_out, int slots, hipStream_t);
hipError_t launch_payload(const float* d_sums, int n, float lin_gain, uint8_t* d_out, hipStream_t);
hipError_t launch_spectra_f64(const SpectraParamsF64&, int in_kind, hipStream_t, int device);
hipError_t launch_welch_accumulate(double* d_acc, const double* d_part, int n, long frames_end, double* d_b, hipStream_t);
hipError_t launch_welch_finish(double* d_acc, int n, long total, double* d_b, hipStream_t);
hipError_t launch_payload_f64(const double* d_sums, int n, double gain, int count, uint8_t* d_out, hipStream_t);

}  // namespace rtlws
#endif
