// frame_input.h -- the input stages of the fused kernels other than the plain cmplx_u8 one,
// written once for both precisions (spectrum_fused.hip: float2 points; spectrum_f64_fused.hip:
// double2 points).  Thread t of a frame's N/16 ends up with x[T*r + t], r < 16, converted as
// the reference converts it BEFORE the 1/128 scale (which the pass-1 twiddles carry):
//   cmplx_s32   (double)s32              src/spectrum.c:74-75
//   real f32    (x, 0)                   src/spectrum.c:92-93
//   CIC-fused   block sum of R consecutive (cmplx_u8 - 128), the value cic_decimate would
//               have written as cmplx_s32 (src/resample.c:24-25,35) -- an integer, exact in
//               either precision
// The window, if any, is the caller's business.
#ifndef RTLWS_FRAME_INPUT_H
#define RTLWS_FRAME_INPUT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtlws_internal.h"
#include "cic_lds.h"

namespace rtlws {

template <typename C2, typename A>
__device__ __forceinline__ C2 mk_point(A x, A y)
{
    typedef decltype(C2::x) R;
    return C2((R)x, (R)y);
}

// P: SpectraParams or SpectraParamsF64 (uses .in and .cic_r); lds: the workgroup's
// transposition buffer, idle at this point of the frame loop (the LDS-staged kinds use it)
template <int N, int IN, typename C2, typename P>
__device__ __forceinline__ void load_frame_points(const P& p, long frame, int t, C2 (&v)[16], void* lds)
{
    constexpr int T = N / 16;
    if constexpr (IN == IN_CS32) {
        const int2* src = reinterpret_cast<const int2*>(p.in) + frame * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int2 s = src[T * r + t];
            v[r] = mk_point<C2>(s.x, s.y);
        }
    } else if constexpr (IN == IN_RF32) {
        const float* src = reinterpret_cast<const float*>(p.in) + frame * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mk_point<C2>(src[T * r + t], 0.0f);
    } else if constexpr (IN == IN_CU8_CIC8) {
        // 8 consecutive cmplx_u8 = one 16-byte load = one decimated sample
        // (reference src/resample.c:24-25,35: block sum of (x - 128)).
        typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
        const nt_u4* src = reinterpret_cast<const nt_u4*>(p.in) + frame * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const nt_u4 s = __builtin_nontemporal_load(src + T * r + t);   // streamed once
            unsigned si = 0, sq = 0;
            si = __builtin_amdgcn_udot4(s.x, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(s.x, 0x01000100u, sq, false);
            si = __builtin_amdgcn_udot4(s.y, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(s.y, 0x01000100u, sq, false);
            si = __builtin_amdgcn_udot4(s.z, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(s.z, 0x01000100u, sq, false);
            si = __builtin_amdgcn_udot4(s.w, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(s.w, 0x01000100u, sq, false);
            v[r] = mk_point<C2>((int)si - 8 * 128, (int)sq - 8 * 128);
        }
    } else if constexpr (IN >= IN_CU8_CICR_LDS4) {
        // R <= cicr_lds_max_r(G), staged through LDS (cic_lds.h).  For one r the
        // 64 samples of a wavefront are one piece (128R contiguous bytes); G
        // pieces are in flight per round.  The slice is the transposition buffer,
        // idle at this point of the frame loop.
        constexpr int G = cicr_lds_round(IN);
        constexpr int RC = cic_ct_factor(IN);          // 10, 12, or 0: run-time factor
        const int R = RC ? RC : p.cic_r;
        const int lane = t & 63;
        const int w = __builtin_amdgcn_readfirstlane(t >> 6);
        const int chunk = 128 * R;
        uint8_t* stage = reinterpret_cast<uint8_t*>(lds) + w * cic_stage_wave_bytes(IN);
        const uint8_t* src = reinterpret_cast<const uint8_t*>(p.in) + (frame * N + 64 * w) * R * 2;
        const long rstride = (long)T * R * 2;
        const CicLaneSum ls = cic_lane_setup(R, lane);
        if constexpr (T > 64) __syncthreads();           // other wavefronts' pass-3 reads are done
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += G) {
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                if constexpr (RC != 0) cic_piece_to_lds_ct<RC>(src + (r0 + rr) * rstride, stage + rr * chunk, lane);
                else cic_piece_to_lds(src + (r0 + rr) * rstride, stage + rr * chunk, R, lane);
            }
            cic_wait_pieces();
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                int2 sum;
                if constexpr (RC != 0) sum = cic_lane_sum_ct<RC>(stage + rr * chunk, lane);
                else sum = cic_lane_sum(stage + rr * chunk, ls);
                v[r0 + rr] = mk_point<C2>(sum.x, sum.y);
                // dword-granular runs (R = 10): keep at most two pieces' worth of LDS
                // loads in registers at a time (eight hoisted pieces = 40 VGPRs spill
                // the windowed K > 1 instantiation)
                if constexpr (RC != 0 && RC % 4 != 0) {
                    if ((rr & 1) == 1) asm volatile("" ::: "memory");
                }
            }
            cic_release_slice();
        }
    } else if constexpr (IN >= IN_CU8_CICR2 && IN <= IN_CU8_CICR16) {
        // any R: a decimated sample is 2R contiguous bytes, read with the widest
        // access its alignment allows (16 B when 8 | R, 8 B when 4 | R, 4 B when
        // 2 | R, else 2 B -- chosen on the host, compile-time here) and reduced
        // with v_dot4_u32_u8 like the R = 8 path.  The sixteen samples of a
        // thread advance together so that sixteen loads are in flight at a time.
        constexpr int W = (IN == IN_CU8_CICR16) ? 16 : (IN == IN_CU8_CICR8) ? 8 : (IN == IN_CU8_CICR4) ? 4 : 2;
        const int R = p.cic_r;
        const int steps = 2 * R / W;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(p.in) + frame * N * R * 2 + (long)t * R * 2;
        const long rstride = (long)T * R * 2;
        unsigned si[16], sq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) si[r] = sq[r] = 0u;
        for (int i = 0; i < steps; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint8_t* q = src + r * rstride + (long)i * W;
                if constexpr (W == 16) {
                    const uint4 s = *reinterpret_cast<const uint4*>(q);
                    si[r] = __builtin_amdgcn_udot4(s.x, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.x, 0x01000100u, sq[r], false);
                    si[r] = __builtin_amdgcn_udot4(s.y, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.y, 0x01000100u, sq[r], false);
                    si[r] = __builtin_amdgcn_udot4(s.z, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.z, 0x01000100u, sq[r], false);
                    si[r] = __builtin_amdgcn_udot4(s.w, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.w, 0x01000100u, sq[r], false);
                } else if constexpr (W == 8) {
                    const uint2 s = *reinterpret_cast<const uint2*>(q);
                    si[r] = __builtin_amdgcn_udot4(s.x, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.x, 0x01000100u, sq[r], false);
                    si[r] = __builtin_amdgcn_udot4(s.y, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s.y, 0x01000100u, sq[r], false);
                } else if constexpr (W == 4) {
                    const unsigned s = *reinterpret_cast<const unsigned*>(q);
                    si[r] = __builtin_amdgcn_udot4(s, 0x00010001u, si[r], false);
                    sq[r] = __builtin_amdgcn_udot4(s, 0x01000100u, sq[r], false);
                } else {
                    const unsigned s = *reinterpret_cast<const uint16_t*>(q);
                    si[r] += s & 0xffu;
                    sq[r] += s >> 8;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = mk_point<C2>((int)si[r] - 128 * R, (int)sq[r] - 128 * R);
    }
}

}  // namespace rtlws
#endif
