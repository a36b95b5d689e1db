// spectrum_f64_4096z.hip -- 4096-point cmplx_u8 frames -> power spectra in DOUBLE for the windowed / K-frame rows
// (BASELINE configs[2] in the reference's arithmetic: src/spectrum.c:54-60 convert, :21 f64 forward DFT, :23-34
// |X|^2 + fft-shift + accumulate + DC-slot rule; K loop of src/cbb_main.c:50-59; dB / payload epilogue of
// src/cbb_main.c:121-130 in double; the Hann window is this build's extension).
//
// Why (round 6; profiles/r06_f64_4096_diag.txt, r06_overlap_probe.txt): a 4096-point frame is four wavefronts; its two
// exchanges move 2 x 64 KiB through the LDS, whose store path takes 13 clocks per ds_write_b128 for the WHOLE CU --
// 2 200 CU-clocks per frame -- beside 3 300 clocks of f64 issue per SIMD.  spectrum_f64_fused.hip at N = 4096 put two
// independent workgroups on a CU and hoped that one's exchanges would fall under the other's butterflies; measured,
// they do not: two workgroups per CU need 8 000 clocks per frame, ONE needs 7 000 (the sum of its own phases), because
// two identical loops that queue for one serial resource fall into step -- both compute, then both wait for the LDS.
// This kernel makes the phase relation a property of the program:
//
//   one workgroup of EIGHT wavefronts per CU = two teams of four, each team a frame of its own; a frame is
//       A  samples -> convert -> window -> pass 1                    (vector ALU)
//       B  exchange 1: the team's only cross-wavefront exchange       (LDS stores)
//       C  exchange-1 reads, pass 2, exchange 2 (wavefront-private), pass 3, |X|^2     (both, mostly ALU)
//   with a barrier between B and C and between C and the next B.  The teams run the SAME loop one barrier apart:
//       team 0:   A B | C | A B | C | ...
//       team 1:       | A B | C | A B | ...
//   so in every interval one team transforms (C) while the other converts and stores (A B): per interval and SIMD
//   2 700 clocks of f64 issue beside 2 200 CU-clocks of LDS traffic, neither waiting for the other by construction.
//
// The transform itself (n = 256 r + 16 r2 + m2, k = q1 + 16 q2 + 256 q3):
//   pass 1  thread t = 16 r2 + m2 of the team's 256 holds x[256 r + t], r < 16: plain radix-16, NO twiddle multiply.
//   exchange 1  (q1, t) -> row q1 of the team's buffer; wavefront w then reads rows 4w .. 4w+3 only: its own quarter.
//   pass 2  lane (q1, m2): radix-16 over r2 with the geometric part (W_256^q1)^r2 of the twiddle owed absorbed
//           (fft_regs_impl.h "last pass", 8 (cos, tan) pairs per lane); the lane-dependent rest is deferred.
//   exchange 2  inside the wavefront, through its own quarter (rows of 16 padded to 17: the conflict-free image of
//           spectrum_f64_1024x.hip): no barrier, wavefront-scope fences only.
//   pass 3  radix-16 over m2: what is owed now, W_4096^(m2 q1) W_256^(m2 q2) = (W_4096^(q1 + 16 q2))^m2, is ONE
//           geometric sequence with a lane-constant ratio, absorbed the same way: no complex twiddle multiplication
//           anywhere, 148 + 192 + 192 = 532 f64 operations per 16 points instead of 148 + 62 + 148 + 192 = 550.
//   samples every wavefront copies ITS sixteen 128-byte pieces of the next frame into a 2 KiB LDS buffer of its own
//           (two global_load_lds_dwordx4: LDS-DMA, no registers) while the current frame is transformed.
//   rows    lane (q2, q1) of wavefront w ends with bins k = q1 + 16 q2 + 256 q3, q3 over its sixteen registers:
//           four neighbouring lanes store four neighbouring bins, the four wavefronts of the team complete every
//           64-byte run between them (plain stores: the L2 merges them), once per K frames.
// The 1/128 input scale rides on nothing: samples enter as (x - 128) w, the power sums are scaled by 2^-14 (exact)
// once per row.  Results equal spectrum_f64_fused.hip's to rounding; tests/test_f64_4096z_gpu.py holds both against
// the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rtlws_internal.h"
#include "fft_regs_f64.h"

namespace rtlws {

using namespace f64;

constexpr int Z_ROW = 272;                       // exchange-1 row: 256 elements padded to 17 x 16 (double2 units)
constexpr int Z_ELEMS = 16 * Z_ROW;              // 4 352 double2 = 69 632 B per team; wavefront w's quarter: [w * 1088, (w + 1) * 1088)
constexpr int Z_TEAM_ELEMS = Z_ELEMS + 1;        // + the team's DC hand-over slot
constexpr int Z_RAW_BYTES = 8 * 2048;            // the eight wavefronts' raw-sample buffers, FIRST in the workgroup's LDS (M0 carries a 16-bit address)
constexpr size_t Z_LDS_BYTES = (size_t)Z_RAW_BYTES + 2 * (size_t)Z_TEAM_ELEMS * 16;      // 155 680 B: one workgroup per CU

size_t spectra_f64_4096z_lds_bytes() { return Z_LDS_BYTES; }

// lane index rebuilt where it is needed (two instructions) instead of held in a register through the frame loop,
// where every register is taken: the allocator spilt such values
__device__ __forceinline__ int z_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wavefront w of a team: its sixteen 128-byte pieces of `frame` -- samples 256 r + 64 w .. + 63, r < 16 -- into its raw
// buffer as [r][64 samples]: lane L of copy j moves 16 bytes of piece r = 8 j + L / 8.  Inline asm on purpose (as in
// spectrum_fused.hip): hipcc would order every later LDS access of the workgroup's one shared array behind a
// __builtin_amdgcn_global_load_lds with vmcnt(0).  The ds_read_u16 of the CURRENT frame are waited for first
// (lgkmcnt(0)): the copies overwrite what they read.  M0 is saved and restored in the same statement.
__device__ __forceinline__ void z_dma_raw(const SpectraParamsF64& p, long frame, int w, unsigned lds_byte_addr)
{
    const int l = z_lane();
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

template <bool WIN, int OUT, bool ROWF32>
__global__ __launch_bounds__(512, 2) void spectra_f64_4096z(const SpectraParamsF64 p)
{
    constexpr int N = 4096;
    static_assert(!(ROWF32 && OUT == OUT_PAYLOAD), "payload rows are bytes in either form");
    extern __shared__ __attribute__((aligned(16))) double2 lds_all[];

    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);        // wave-uniform: scalar registers
    const int w = __builtin_amdgcn_readfirstlane(((int)threadIdx.x >> 6) & 3);     // wavefront within the team
    const int tid = threadIdx.x & 255;            // thread within the team: t = 16 r2 + m2 in pass 1
    const int l = tid & 63;
    const int K = p.k_avg;
    const long ngroups = p.ngroups;
    double2* const ldsd = lds_all + Z_RAW_BYTES / 16 + team * Z_TEAM_ELEMS;        // the team's exchange buffer
    double2* const slice = ldsd + w * (4 * Z_ROW);                                  // this wavefront's quarter of it
    double* const dc_slot = reinterpret_cast<double*>(ldsd + Z_ELEMS);

    // The team's rows: g0 + i * stride.  Team 0 never has fewer rows than team 1; the loop below runs the barriers of
    // the team with the most frames and every wavefront of the workgroup executes every one of them.
    const long stride = 2L * gridDim.x;
    const long g0 = 2L * blockIdx.x + team;
    const long my_rows = g0 < ngroups ? (ngroups - g0 + stride - 1) / stride : 0;
    const long my_frames = my_rows * K;
    const long lead_rows = 2L * blockIdx.x < ngroups ? (ngroups - 2L * blockIdx.x + stride - 1) / stride : 0;
    const long last_slot = 2 * lead_rows * K + 1;           // team 1's epilogue slot when it has as many frames as team 0

    // this wavefront's raw-sample buffer (2 KiB) and its LDS byte address (for M0); the first frame's copy goes out
    // before anything else
    const int wave8 = 4 * team + w;
    const uint16_t* const rawl = reinterpret_cast<const uint16_t*>(lds_all) + wave8 * 1024 + l;
    const unsigned raw_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(reinterpret_cast<uint8_t*>(lds_all) + wave8 * 2048);
    if (my_frames > 0) z_dma_raw(p, g0 * K, w, raw_addr);

    // lane constants, resident for the life of the (persistent) workgroup: pass 2's pairs of alpha = W_256^q1
    // (lane (q1, m2) = (tid >> 4, tid & 15)), pass 3's of beta = W_4096^(q1 + 16 q2) (lane (q2, q1): the host
    // lays the table out by thread), the Hann weights from two lane constants
    f2 twA[8], twB[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) twA[m] = p.tw2f[(tid >> 4) * 8 + m];
#pragma unroll
    for (int m = 0; m < 8; ++m) twB[m] = p.twyb[tid * 8 + m];
    const f2 wcs = WIN ? p.hann_csf[tid] : mk(0.0, 0.0);
    // (eight weights, not sixteen -- which this kernel has no registers for: w_(r+8) = 1 - w_r, so the second half
    // of the samples enters as x - x w_r, one FMA, one rounding of the exact value)
    double win[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) win[r] = WIN ? hann_w(r, wcs) : 1.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(twA[m].x), "v"(twA[m].y), "v"(twB[m].x), "v"(twB[m].y));
    if constexpr (WIN) {
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(win[r]));
    }

    const int wp = l >> 4, wc = l & 15;           // exchange 2, writer side: lane (p, c) = (q1 & 3, m2)

    double acc[16];
    double wdc = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0;

    // ---- the row of K frames that ended in the team's previous slot: DC-slot rule (src/spectrum.c:25-33: slot N/2 --
    // bin 0: wavefront 0, lane 0, slot 0 -- takes sum_k (K-k) P_k[N-1], left in the team's slot by its last thread
    // before the barrier), epilogue, stores; then the accumulators start over
    auto finish_row = [&](long g) {
        // (addresses rebuilt from the lane index: hoisted out of the loop they would be live through the transforms)
        const int lr = z_lane();
        if (w == 0 && lr == 0) acc[0] = *dc_slot;
        // lane (q2, q1) = (l >> 2, 4 w + (l & 3)) holds bins k = q1 + 16 q2 + 256 q3; fft-shift = q3 ^ 8
        const int kk = 4 * w + (lr & 3) + 16 * (lr >> 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const long i = g * N + kk + 256 * (rev16(u) ^ 8);
            const double a = acc[u] * 0x1p-14;              // the (x - 128) scale: (1/128)^2, exact
            if constexpr (OUT == OUT_PAYLOAD) {
                // src/cbb_main.c:125-128, same operation order, in double
                const double d = 10.0 * log10(fabs(p.lin_gain * a / (double)p.count));
                const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
                reinterpret_cast<uint8_t*>(p.out)[i] = (uint8_t)m;
            } else {
                const double o = (OUT == OUT_DB) ? 10.0 * log10(a / (double)p.count) : a;
                if constexpr (ROWF32) reinterpret_cast<float*>(p.out)[i] = (float)o;
                else reinterpret_cast<double*>(p.out)[i] = o;
            }
            acc[u] = 0.0;
        }
        wdc = 0.0;
    };

    // the team's position: frame kf of row `row` (no division in the loop: the counters step)
    long row = 0;
    int kf = 0;
    for (long slot = 0; slot <= last_slot; ++slot) {
        const long mine = slot - team;            // this team's own slot count: even = A B of frame mine / 2, odd = C of it
        const long f = mine >> 1;                 // the team's frame index = row * K + kf
        if (mine >= 0 && mine < 2 * my_frames && !(mine & 1)) {
            // ---- A: this frame's samples have landed in the wavefront's buffer (its copies are the only vector-memory
            // loads in flight; the stores of the row before last are long retired)
            f2 v[16];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned raw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) raw[r] = rawl[64 * r];
            if (f + 1 < my_frames) {                                 // (wave-uniform; waits for the reads above)
                const long nframe = kf + 1 < K ? (g0 + row * stride) * K + kf + 1 : (g0 + (row + 1) * stride) * K;
                z_dma_raw(p, nframe, w, raw_addr);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // rectangular: (double)u8, the 128 offset kept (it only reaches bin 0, which is never output:
                // src/spectrum.c:31); windowed: (x - 128) w with the subtraction on the integers (exact) -- one
                // rounding, the same value as fma(x, w, -128 w), without sixteen more lane constants
                if constexpr (WIN) {
                    const double re = (double)((int)(raw[r] & 0xffu) - 128), im = (double)((int)((raw[r] >> 8) & 0xffu) - 128);
                    v[r] = r < 8 ? mk(re * win[r], im * win[r]) : mk(fma(-re, win[r - 8], re), fma(-im, win[r - 8], im));
                } else {
                    v[r] = mk((double)(raw[r] & 0xffu), (double)((raw[r] >> 8) & 0xffu));
                }
            }
            // pass 1: radix-16 over r; slot s holds q1 = rev16(s)
            fft16_sel(v);
            // ---- B: exchange 1 (the team's four wavefronts): (q1, t) -> row q1.  The barrier before this slot saw
            // every wavefront of the team through its exchange-2 reads of the previous frame.
#pragma unroll
            for (int s = 0; s < 16; ++s) ldsd[rev16(s) * Z_ROW + tid] = v[s];
            // a row ended in the team's previous slot: its epilogue and stores go here -- the samples and the
            // transform's registers are dead, and the next wait for vector memory is two slots away
            if (kf == 0 && f > 0) finish_row(g0 + (row - 1) * stride);
        } else if (mine >= 0 && mine < 2 * my_frames) {
            // ---- C: lane (q1, m2) = (tid >> 4, tid & 15) reads r2 = 0 .. 15: rows 4w .. 4w+3 only
            f2 v[16];
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) v[r2] = ldsd[(tid >> 4) * Z_ROW + 16 * r2 + (tid & 15)];
            // pass 2: radix-16 over r2, (W_256^q1)^r2 absorbed; slot s holds q2 = rev16(s)
            fft_last<16>(v, 0, twA);
            // exchange 2, inside the wavefront and inside its own quarter (which only this wavefront reads):
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
            // pass 3: radix-16 over m2, (W_4096^(q1 + 16 q2))^m2 absorbed; slot s holds q3 = rev16(s)
            fft_last<16>(v, 0, twB);
            // |X|^2, accumulate; bin N-1 (q1 = q2 = q3 = 15: wavefront 3, lane 63, slot 15) also feeds the DC slot
            // with weight K - kf
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
            if (kf == K - 1 && tid == 255) *dc_slot = wdc;      // read by the team's first thread after the next barrier
            if (++kf == K) { kf = 0; ++row; }
        } else if (mine == 2 * my_frames && my_frames > 0) {
            finish_row(g0 + (my_rows - 1) * stride);            // the team's last row
        }
        __syncthreads();
    }
}

template <bool WIN, int OUT, bool ROWF32>
static hipError_t launch_z_one(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    static std::atomic<unsigned long long> ready{0};          // > 64 KiB of LDS: the attribute once per device
    const unsigned long long bit = 1ull << (device & 63);
    if (!(ready.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectra_f64_4096z<WIN, OUT, ROWF32>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z_LDS_BYTES);
        if (e != hipSuccess) return e;
        ready.fetch_or(bit, std::memory_order_release);
    }
    if (blocks <= 0) return hipSuccess;      // rtlws_engine_prepare_f64: the attribute only, nothing enqueued
    hipLaunchKernelGGL((spectra_f64_4096z<WIN, OUT, ROWF32>), dim3(blocks), dim3(512), Z_LDS_BYTES, st, p);
    return hipGetLastError();
}

template <bool WIN>
static hipError_t launch_z_o(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    switch (p.out_mode) {
    case OUT_SUM: return p.rows_f32 ? launch_z_one<WIN, OUT_SUM, true>(p, blocks, st, device)
                                    : launch_z_one<WIN, OUT_SUM, false>(p, blocks, st, device);
    case OUT_DB: return p.rows_f32 ? launch_z_one<WIN, OUT_DB, true>(p, blocks, st, device)
                                   : launch_z_one<WIN, OUT_DB, false>(p, blocks, st, device);
    default: return launch_z_one<WIN, OUT_PAYLOAD, false>(p, blocks, st, device);
    }
}

// cmplx_u8 frames of 4096 points, any K >= 1, Hann or rectangular; `blocks` persistent workgroups of 512 threads
// (one per CU: 152 KiB of LDS each), two rows in flight per workgroup
hipError_t launch_spectra_f64_4096z(const SpectraParamsF64& p, int blocks, hipStream_t st, int device)
{
    return p.window ? launch_z_o<true>(p, blocks, st, device) : launch_z_o<false>(p, blocks, st, device);
}

}  // namespace rtlws
