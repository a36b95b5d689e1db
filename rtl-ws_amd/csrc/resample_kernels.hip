// resample_kernels.hip -- the two decimators of reference src/resample.c as
// stand-alone HIP kernels (the CIC also exists fused into the spectrum kernel).
//
// cic_block_sums: reference src/resample.c:21-40 reduces, once the delay-line
//   bookkeeping is separated out, to dst[m] = sum_{n<R}(src[m*R+n] - 128) per
//   component in int32 -- exact integer work, one pass over the bytes:
//   2R bytes read and 8 bytes written per output.  HBM-bound; R = 8 is one
//   16-byte load and one 8-byte store per thread.
// halfband: reference src/resample.c:53-64, f32, products and sums in source
//   order with no contraction so that results are bit-identical to an IEEE
//   evaluation of the reference's expression.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtlws_internal.h"
#include "cic_lds.h"

namespace rtlws {

typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
typedef int nt_i2 __attribute__((ext_vector_type(2)));

// Streamed once in, once out: nontemporal both ways.
__global__ __launch_bounds__(256) void cic8_kernel(const nt_u4* __restrict__ src,
                                                   nt_i2* __restrict__ dst, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < n; m += stride) {
        const nt_u4 s = __builtin_nontemporal_load(src + m);
        unsigned si = 0, sq = 0;
        si = __builtin_amdgcn_udot4(s.x, 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(s.x, 0x01000100u, sq, false);
        si = __builtin_amdgcn_udot4(s.y, 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(s.y, 0x01000100u, sq, false);
        si = __builtin_amdgcn_udot4(s.z, 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(s.z, 0x01000100u, sq, false);
        si = __builtin_amdgcn_udot4(s.w, 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(s.w, 0x01000100u, sq, false);
        const nt_i2 o = {(int)si - 8 * 128, (int)sq - 8 * 128};
        __builtin_nontemporal_store(o, dst + m);
    }
}

// Any 1 <= R <= 128, staged through LDS (cic_lds.h).  One wavefront owns rounds
// of G consecutive pieces (a piece = 64 outputs = 128R contiguous input bytes),
// all G copies in flight together, then one 8-byte nontemporal store per lane
// and piece.  The < 64 outputs that do not fill a piece are summed straight from
// global memory by one wavefront.
// RC: the factor as a compile-time constant (10, 12: the reference's own,
// src/main.c:23,154 -- unrolled copies and conflict-free unrolled sums), or 0.
template <int RC>
__global__ __launch_bounds__(256) void cicr_kernel(const uint8_t* __restrict__ src,
                                                   nt_i2* __restrict__ dst, long n, int R_rt,
                                                   int slice_bytes, int G)
{
    const int R = RC ? RC : R_rt;
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];   // 4 wavefronts * slice_bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint8_t* my = stage + wave * slice_bytes;
    const int chunk = 128 * R;
    const long npieces = n / 64;
    const long round_stride = (long)gridDim.x * 4 * G;
    const CicLaneSum ls = cic_lane_setup(R, lane);
    for (long p0 = ((long)blockIdx.x * 4 + wave) * G; p0 < npieces; p0 += round_stride) {
        const int gn = (npieces - p0 < G) ? (int)(npieces - p0) : G;
        for (int g = 0; g < gn; ++g) {
            if constexpr (RC != 0) cic_piece_to_lds_ct<RC>(src + (p0 + g) * chunk, my + g * chunk, lane);
            else cic_piece_to_lds(src + (p0 + g) * chunk, my + g * chunk, R, lane);
        }
        cic_wait_pieces();
        for (int g = 0; g < gn; ++g) {
            int2 sum;
            if constexpr (RC != 0) sum = cic_lane_sum_ct<RC>(my + g * chunk, lane);
            else sum = cic_lane_sum(my + g * chunk, ls);
            const nt_i2 o = {sum.x, sum.y};
            __builtin_nontemporal_store(o, dst + (p0 + g) * 64 + lane);
        }
        cic_release_slice();
    }
    const long m = npieces * 64 + lane;
    if (blockIdx.x == 0 && wave == 0 && m < n) {
        const uint16_t* q = reinterpret_cast<const uint16_t*>(src) + m * R;
        unsigned si = 0, sq = 0;
        for (int i = 0; i < R; ++i) {
            const unsigned x = q[i];
            si += x & 0xffu;
            sq += x >> 8;
        }
        const nt_i2 o = {(int)si - 128 * R, (int)sq - 128 * R};
        dst[m] = o;
    }
}

// cus: compute units of the engine's device (the grids are persistent: a few workgroups per CU)
hipError_t launch_cic_block_sums(int R, const void* d_src, long dst_len, void* d_dst,
                                 hipStream_t st, int cus)
{
    if (dst_len <= 0) return hipSuccess;
    if (cus < 1) cus = 1;
    if (R == 8) {
        long blocks = (dst_len + 255) / 256;
        if (blocks > (long)cus * 8) blocks = (long)cus * 8;
        hipLaunchKernelGGL(cic8_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const nt_u4*>(d_src), reinterpret_cast<nt_i2*>(d_dst),
                           dst_len);
    } else {
        if (R < 1 || R > 128) return hipErrorInvalidValue;
        // 8 KiB of LDS per wavefront (16 KiB when a piece is larger): 4 workgroups
        // of 4 wavefronts per CU, 8 KiB in flight per wavefront
        const int chunk = 128 * R;
        const int slice = chunk > 8192 ? 16384 : 8192;
        const int G = slice / chunk;
        const long rounds = (dst_len / 64 + G - 1) / G;
        long blocks = (rounds + 3) / 4;
        const long cap = (long)cus * (slice == 8192 ? 4 : 2);
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        const uint8_t* s8 = reinterpret_cast<const uint8_t*>(d_src);
        nt_i2* d2 = reinterpret_cast<nt_i2*>(d_dst);
        const dim3 grid((unsigned)blocks), block(256);
        const size_t lds = (size_t)4 * slice;
        if (R == 10) hipLaunchKernelGGL(cicr_kernel<10>, grid, block, lds, st, s8, d2, dst_len, R, slice, G);
        else if (R == 12) hipLaunchKernelGGL(cicr_kernel<12>, grid, block, lds, st, s8, d2, dst_len, R, slice, G);
        else hipLaunchKernelGGL(cicr_kernel<0>, grid, block, lds, st, s8, d2, dst_len, R, slice, G);
    }
    return hipGetLastError();
}

// atan2_approx of reference src/common_sp.h:40-76, evaluated as the C source
// reads under IEEE rules: f32 divide/multiply/add without contraction (this
// file is built with -ffp-contract=off), and the +-M_PI corrections as a
// double-precision add rounded back to f32.
__device__ __forceinline__ float atan2_approx_dev(float y, float x)
{
    const float pi_by_2 = (float)(3.14159265358979323846 / 2);
    const double pi_d = 3.14159265358979323846;
    if (x == 0.0f) {
        if (y > 0.0f) return pi_by_2;
        if (y == 0.0f) return 0.0f;
        return -pi_by_2;
    }
    const float z = __fdiv_rn(y, x);
    if (fabsf(z) < 1.0f) {
        const float at = __fdiv_rn(z, __fadd_rn(1.0f, __fmul_rn(__fmul_rn(0.28f, z), z)));
        if (x < 0.0f) {
            if (y < 0.0f) return (float)((double)at - pi_d);
            return (float)((double)at + pi_d);
        }
        return at;
    }
    const float at = __fsub_rn(pi_by_2, __fdiv_rn(z, __fadd_rn(__fmul_rn(z, z), 0.28f)));
    if (y < 0.0f) return (float)((double)at - pi_d);
    return at;
}

// reference src/audio_main.c:110-131: phase, first difference, hard limit.
// Each thread recomputes its left neighbour's phase (elementwise, no scan).
__global__ __launch_bounds__(256) void fm_demod_kernel(const int2* __restrict__ iq, long n,
                                                       const float* __restrict__ prev_in,
                                                       float* __restrict__ prev_out,
                                                       float* __restrict__ out)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int2 s = iq[i];
        const float ph = atan2_approx_dev((float)s.y, (float)s.x);
        float pp;
        if (i == 0) {
            pp = *prev_in;
        } else {
            const int2 q = iq[i - 1];
            pp = atan2_approx_dev((float)q.y, (float)q.x);
        }
        float d = __fsub_rn(ph, pp);
        if (d > 1.0f) d = 1.0f;
        else if (d < -1.0f) d = -1.0f;
        out[i] = d;
        if (i == n - 1) *prev_out = ph;
    }
}

hipError_t launch_fm_demod(const void* d_iq, long len, const float* d_prev_in, float* d_prev_out,
                           float* d_out, hipStream_t st, int cus)
{
    if (len <= 0) return hipSuccess;
    if (cus < 1) cus = 1;
    long blocks = (len + 255) / 256;
    if (blocks > (long)cus * 8) blocks = (long)cus * 8;
    hipLaunchKernelGGL(fm_demod_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const int2*>(d_iq), len, d_prev_in, d_prev_out, d_out);
    return hipGetLastError();
}

// x points 10 floats into the buffer, so x[-10..-1] is the delay line
// (reference src/resample.c:57,63: delay[(HALF_BAND_N - 1) + idx]).
__global__ __launch_bounds__(256) void halfband_kernel(const float* __restrict__ xbuf,
                                                       float* __restrict__ y, long n)
{
    // This file is compiled with -ffp-contract=off (Makefile): mul then add, never
    // fma, for bit parity with an IEEE evaluation of the C expression.
    const float h0 = 0.01824f, h2 = -0.11614f, h4 = 0.34790f, h5 = 0.5f;   // src/resample.c:4
    const float* x = xbuf + 10;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* c = x + 2 * i;
        float acc = __fmul_rn(h5, c[-5]);                    // src/resample.c:57
        acc = __fadd_rn(acc, __fmul_rn(h0, c[0]));           // k = 0   src/resample.c:60-64
        acc = __fadd_rn(acc, __fmul_rn(h2, c[-2]));          // k = 2
        acc = __fadd_rn(acc, __fmul_rn(h4, c[-4]));          // k = 4
        acc = __fadd_rn(acc, __fmul_rn(h4, c[-6]));          // k = 6
        acc = __fadd_rn(acc, __fmul_rn(h2, c[-8]));          // k = 8
        acc = __fadd_rn(acc, __fmul_rn(h0, c[-10]));         // k = 10
        y[i] = acc;
    }
}

hipError_t launch_halfband(const float* d_x, float* d_y, long out_len, hipStream_t st, int cus)
{
    if (out_len <= 0) return hipSuccess;
    if (cus < 1) cus = 1;
    long blocks = (out_len + 255) / 256;
    if (blocks > (long)cus * 8) blocks = (long)cus * 8;
    hipLaunchKernelGGL(halfband_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_x, d_y, out_len);
    return hipGetLastError();
}

// ---- shader-clock probe (measurement infrastructure; include/rtlws_hip.h, rtlws_clock_probe_*) ----
// ONE wavefront that sits beside the kernels being timed and reads the two hardware counters at its
// start and when told to stop: s_memtime counts shader clocks, s_memrealtime a fixed 100 MHz
// (MI355X_MICROARCH.md, DVFS): d(memtime) / d(memrealtime) x 100 MHz is the clock the chip actually
// ran at over that interval -- the interval of the timed launches, not of another launch series.
// It sleeps between polls (s_sleep: no issue slots, no memory traffic but one 4-byte read per ~0.5 us)
// and ALWAYS terminates: on the stop flag, after max_ticks of the 100 MHz counter (a bound in TIME: a poll is a
// sleep plus a system-scope load whose latency depends on what else runs), or after max_polls polls.
__global__ __launch_bounds__(64) void clock_probe_kernel(const int* stop, unsigned long long* out,
                                                         int max_polls, unsigned long long max_ticks)
{
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    if (threadIdx.x == 0) {                 // tells the host it is resident: the caller's clock starts after this
        __hip_atomic_store(&out[3], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int polls = 0;
    while (polls < max_polls) {
        __builtin_amdgcn_s_sleep(16);       // ~0.5 us between polls
        ++polls;
        if (__hip_atomic_load(stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) break;   // past every cache
        if (wall_clock64() - r0 > max_ticks) break;
    }
    if (threadIdx.x == 0) {
        out[0] = clock64() - c0;
        out[1] = wall_clock64() - r0;
        out[2] = (unsigned long long)polls;
    }
}

// ---- shader-clock stamp (measurement infrastructure; include/rtlws_hip.h, rtlws_clock_stamp) ----
// `slots` one-wavefront workgroups that each write the two hardware counters and WHERE they ran, and leave.  Two such
// launches in one stream, around the launches being measured, bracket them with NOTHING resident beside them (the
// probe above takes registers and a second hardware queue: round 6 measured what that costs the launches it sits
// beside).  s_memtime is a counter of the place it is read at (two one-wavefront stamps on different CUs differ by
// arbitrary offsets), so the host pairs the records of the two launches BY PLACE -- XCC, SE, SH, CU, SIMD -- and
// takes d(memtime) / d(memrealtime) per place.
__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* out)
{
    if (threadIdx.x == 0) {
        unsigned long long* o = out + 4 * (size_t)blockIdx.x;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u; // XCC_ID
        o[0] = clock64();                                            // s_memtime: shader clocks
        o[1] = wall_clock64();                                       // s_memrealtime: 100 MHz
        o[2] = ((unsigned long long)xcc << 16) | (hw & 0xff30u);     // the place
        o[3] = 0x5354414d50ull;                                      // written
    }
}

hipError_t launch_clock_stamp(unsigned long long* d_out, int slots, hipStream_t st)
{
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(slots), dim3(64), 0, st, d_out);
    return hipGetLastError();
}

hipError_t launch_clock_probe(const int* stop_flag, unsigned long long* out, int max_polls, hipStream_t st)
{
    // 10 s of the 100 MHz counter, whatever a poll costs
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, stop_flag, out, max_polls, 1000000000ull);
    return hipGetLastError();
}

}  // namespace rtlws
