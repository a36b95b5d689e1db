// cic_lds.h -- CIC block sums of cmplx_u8 through LDS-DMA staging; shared by the
// fused spectrum kernel's input stage (spectrum_fused.hip) and the stand-alone
// decimator (resample_kernels.hip).
//
// Reference src/resample.c:21-40 reduces to dst[m] = sum_{n<R}(src[m*R+n] - 128)
// per component.  A "piece" is the input of 64 consecutive outputs: 128R
// contiguous bytes, 128-byte aligned.  A wavefront copies whole pieces to its
// own LDS slice with global_load_lds (whole cache lines whatever R is, no VGPRs,
// any number in flight) and each lane then sums its own 2R bytes from LDS with
// v_dot4_u32_u8.  An integer sum: the order of the terms is free.
#ifndef RTLWS_CIC_LDS_H
#define RTLWS_CIC_LDS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtlws {

// cache policy of the staging loads: 2 = nontemporal (streamed once; measured
// +8-10 % over the default policy, tools/cic_fused_rates.py)
#define RTLWS_GLDS_AUX 2

// Issue the copy of one piece (128R bytes at s) to LDS at d (wave-uniform, 16-byte
// aligned): 1 KiB pieces, 256-byte pieces and, for odd R, one 128-byte piece by
// the lower half of the wavefront.  Asynchronous: retire with s_waitcnt vmcnt(0).
__device__ __forceinline__ void cic_piece_to_lds(const uint8_t* s, uint8_t* d, int R, int lane)
{
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int chunk = 128 * R;
    const int n16 = chunk >> 10, n4 = (chunk & 1023) >> 8, tail = chunk & 128;
    for (int i = 0; i < n16; ++i)
        __builtin_amdgcn_global_load_lds((glb_vp)(s + i * 1024 + lane * 16), (lds_vp)(d + i * 1024), 16, 0, RTLWS_GLDS_AUX);
    s += n16 * 1024;
    d += n16 * 1024;
    for (int i = 0; i < n4; ++i)
        __builtin_amdgcn_global_load_lds((glb_vp)(s + i * 256 + lane * 4), (lds_vp)(d + i * 256), 4, 0, RTLWS_GLDS_AUX);
    if (tail && lane < 32)
        __builtin_amdgcn_global_load_lds((glb_vp)(s + n4 * 256 + lane * 4), (lds_vp)(d + n4 * 256), 4, 0, RTLWS_GLDS_AUX);
}

__device__ __forceinline__ void cic_wait_pieces()   // an LDS-DMA is a pending LDS write on vmcnt
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void cic_release_slice() // LDS reads retired before the slice is overwritten
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// A lane's run of 2R bytes read as S dwords from the dword at or below its first
// byte.  Odd R: runs start on alternating halves of a dword, so one dword (the
// first of an odd lane, the last of an even one) is half the neighbour's and is
// masked.  Even R: the lane stride is S dwords; when S is even each lane starts
// its sum at a different dword so that the lanes spread over the banks.
struct CicLaneSum {
    int S, lane_off, jmask, idx0, bias;
    unsigned hmask;
};

__device__ __forceinline__ CicLaneSum cic_lane_setup(int R, int lane)
{
    CicLaneSum c;
    c.S = (R + 1) >> 1;
    c.lane_off = (lane * 2 * R) & ~3;
    c.jmask = (R & 1) ? ((lane & 1) ? 0 : c.S - 1) : -1;
    c.hmask = (lane & 1) ? 0xffff0000u : 0x0000ffffu;
    c.idx0 = ((R & 1) || (c.S & 1)) ? 0 : (lane * c.S / 32) % c.S;
    c.bias = 128 * R;
    return c;
}

// sum of (x - 128) over this lane's run of the staged piece: re -> .x, im -> .y
__device__ __forceinline__ int2 cic_lane_sum(const uint8_t* piece, const CicLaneSum& c)
{
    const unsigned* q = reinterpret_cast<const unsigned*>(piece + c.lane_off);
    unsigned si = 0u, sq = 0u;
    int idx = c.idx0;
    for (int j = 0; j < c.S; ++j) {
        unsigned x = q[idx];
        if (idx == c.jmask) x &= c.hmask;
        if (++idx == c.S) idx = 0;
        si = __builtin_amdgcn_udot4(x, 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(x, 0x01000100u, sq, false);
    }
    return make_int2((int)si - c.bias, (int)sq - c.bias);
}

// Compile-time factor (the reference's own: sample_rate / 192000 = 10 at 2.048 MS/s,
// 12 at 2.4 MS/s, src/main.c:23,154): the run of 2*RC bytes is read with fully
// unrolled, conflict-free LDS loads and no scalar loop, no mask, no rotation.
//   RC % 4 == 0: runs are 8-byte aligned -> RC/4 ds_read_b64; within a 32-lane
//     group the lanes start at dword (RC/2)*l, and for RC = 12 the pairs
//     {6l, 6l+1} mod 64 tile the 64 banks exactly once (gcd(3, 32) = 1).
//   else (RC even): RC/2 ds_read_b32 at a lane stride of RC/2 dwords, odd for
//     RC = 10 (5l mod 32 is a permutation of the banks).
template <int RC>
__device__ __forceinline__ int2 cic_lane_sum_ct(const uint8_t* piece, int lane)
{
    static_assert(RC >= 2 && RC % 2 == 0, "compile-time CIC factors are even");
    unsigned si = 0u, sq = 0u;
    if constexpr (RC % 4 == 0) {
        const uint2* q = reinterpret_cast<const uint2*>(piece + lane * (2 * RC));
#pragma unroll
        for (int j = 0; j < RC / 4; ++j) {
            const uint2 x = q[j];
            si = __builtin_amdgcn_udot4(x.x, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(x.x, 0x01000100u, sq, false);
            si = __builtin_amdgcn_udot4(x.y, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(x.y, 0x01000100u, sq, false);
        }
    } else {
        const unsigned* q = reinterpret_cast<const unsigned*>(piece + lane * (2 * RC));
#pragma unroll
        for (int j = 0; j < RC / 2; ++j) {
            const unsigned x = q[j];
            si = __builtin_amdgcn_udot4(x, 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(x, 0x01000100u, sq, false);
        }
    }
    return make_int2((int)si - 128 * RC, (int)sq - 128 * RC);
}

// cic_piece_to_lds with the factor known at compile time: the piece is
// (128*RC) / 1024 1-KiB copies plus 256-byte copies, all unrolled.  One per-lane
// address per access width; the position inside the piece is the instruction's
// immediate offset, which the hardware adds to BOTH the global and the LDS
// address (LDS = M0 base + offset + lane * width) -- no address arithmetic and
// no M0 rewrite per copy.
template <int OFF0, int I, int COUNT>
__device__ __forceinline__ void cic_copies16_ct(const uint8_t* g, uint8_t* d)    // 1 KiB per copy
{
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    if constexpr (I < COUNT) {
        __builtin_amdgcn_global_load_lds((glb_vp)g, (lds_vp)d, 16, OFF0 + I * 1024, RTLWS_GLDS_AUX);
        cic_copies16_ct<OFF0, I + 1, COUNT>(g, d);
    }
}
template <int OFF0, int I, int COUNT>
__device__ __forceinline__ void cic_copies4_ct(const uint8_t* g, uint8_t* d)     // 256 B per copy
{
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    if constexpr (I < COUNT) {
        __builtin_amdgcn_global_load_lds((glb_vp)g, (lds_vp)d, 4, OFF0 + I * 256, RTLWS_GLDS_AUX);
        cic_copies4_ct<OFF0, I + 1, COUNT>(g, d);
    }
}

// (Measured and not kept, profiles/r03_ab_cic_tail_x4.txt: the sub-KiB tail of a piece as ONE 16-byte-per-lane
// copy by the first 32 / 16 lanes instead of two / one 4-byte-per-lane copies by all 64;
// tools/variants/csrc_hooks.patch restores the switch.)

template <int RC>
__device__ __forceinline__ void cic_piece_to_lds_ct(const uint8_t* s, uint8_t* d, int lane)
{
    constexpr int chunk = 128 * RC;
    constexpr int n16 = chunk >> 10, n4 = (chunk & 1023) >> 8;
    static_assert((chunk & 255) == 0, "even factors only");
    cic_copies16_ct<0, 0, n16>(s + lane * 16, d);
    cic_copies4_ct<n16 * 1024, 0, n4>(s + lane * 4, d);
}

}  // namespace rtlws
#endif
