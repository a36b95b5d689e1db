// spectrum_direct.hip -- direct-DFT power-spectrum kernel for any N >= 2.
//
// Completeness path behind spectrum_alloc(N) for sizes the fused kernel does
// not cover (the reference accepts any N because FFTW does,
// reference src/spectrum.c:37-45).  O(N^2), f32; same conversion, fft-shift,
// DC-slot rule, K-frame accumulation and epilogues as the fused kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtlws_internal.h"

namespace rtlws {

typedef float2 f2;
__device__ __forceinline__ f2 mk(float x, float y) { return make_float2(x, y); }

// ---- direct DFT kernel: any N >= 2 (slow, complete) -------------------------
// One workgroup per output row; thread b computes bins b, b+256, ... from a
// frame staged in LDS as f32 pairs, with a W_N table in global memory.
template <int IN>
__global__ __launch_bounds__(256) void spectra_direct(const SpectraParams p)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int N = p.n_fft;
    const int K = p.k_avg;
    const int R = p.cic_r > 1 ? p.cic_r : 1;
    const long g = blockIdx.x;
    for (int b0 = 0; b0 < N; b0 += blockDim.x) {
        const int i = b0 + threadIdx.x;            // output slot
        const int k = (i + N / 2) % N;              // bin this slot shows (src/spectrum.c:25)
        float sum = 0.0f, sum_nb = 0.0f;           // own bin, and bin N-1 weights for the DC slot
        for (int kf = 0; kf < K; ++kf) {
            const long frame = g * K + kf;
            __syncthreads();
            for (int n = threadIdx.x; n < N; n += blockDim.x) {
                float re, im;
                if (IN == IN_CS32) {
                    const int2 s = reinterpret_cast<const int2*>(p.in)[frame * N + n];
                    re = (float)s.x; im = (float)s.y;
                } else if (IN == IN_RF32) {
                    re = reinterpret_cast<const float*>(p.in)[frame * N + n]; im = 0.0f;
                } else {
                    const uint8_t* q = reinterpret_cast<const uint8_t*>(p.in) + ((frame * N + n) * (long)R) * 2;
                    int si = 0, sq = 0;
                    for (int r = 0; r < R; ++r) { si += q[2 * r] - 128; sq += q[2 * r + 1] - 128; }
                    re = (float)si; im = (float)sq;
                }
                if (p.window) { re *= p.window[n]; im *= p.window[n]; }
                lds[n] = mk(re * p.in_scale, im * p.in_scale);
            }
            __syncthreads();
            if (i < N) {
                const int kk = (k == 0) ? N - 1 : k;   // DC slot mirrors bin N-1
                float ar = 0.0f, ai = 0.0f;
                int e = 0;
                for (int n = 0; n < N; ++n) {
                    const f2 w = p.tw1[e];
                    const f2 x = lds[n];
                    ar = fmaf(x.x, w.x, fmaf(-x.y, w.y, ar));
                    ai = fmaf(x.x, w.y, fmaf(x.y, w.x, ai));
                    e += kk; if (e >= N) e -= N;
                }
                const float pw = ar * ar + ai * ai;
                sum += pw;
                sum_nb = fmaf((float)(K - kf), pw, sum_nb);
            }
        }
        if (i < N) {
            float a = (k == 0) ? sum_nb : sum;
            if (p.out_mode == OUT_DB) a = fmaf(10.0f, log10f(a), p.db_offset);
            if (p.out_mode == OUT_PAYLOAD) {
                const float d = 10.0f * log10f(fabsf(a * p.lin_gain));
                const unsigned m = (d >= 0.0f) ? (d <= 255.0f ? (unsigned)(int)d : 255u) : 0u;
                reinterpret_cast<uint8_t*>(p.out)[g * N + i] = (uint8_t)m;
            } else {
                reinterpret_cast<float*>(p.out)[g * N + i] = a;
            }
        }
    }
}


// dB / truncate / clamp of reference src/cbb_main.c:125-128 as its own kernel.
__global__ __launch_bounds__(256) void payload_kernel(const float* __restrict__ sums, int n,
                                                      float lin_gain, uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float d = 10.0f * log10f(fabsf(sums[i] * lin_gain));
        const unsigned m = (d >= 0.0f) ? (d <= 255.0f ? (unsigned)(int)d : 255u) : 0u;
        out[i] = (uint8_t)m;
    }
}

hipError_t launch_payload(const float* d_sums, int n, float lin_gain, uint8_t* d_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(payload_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_sums, n, lin_gain, d_out);
    return hipGetLastError();
}

hipError_t launch_spectra_direct(const SpectraParams& p, int in_kind, hipStream_t st)
{
    const size_t lds_bytes = sizeof(float2) * (size_t)p.n_fft;
    const dim3 grid((unsigned)p.ngroups), block(256);
    if (in_kind == IN_CS32)
        hipLaunchKernelGGL((spectra_direct<IN_CS32>), grid, block, lds_bytes, st, p);
    else if (in_kind == IN_RF32)
        hipLaunchKernelGGL((spectra_direct<IN_RF32>), grid, block, lds_bytes, st, p);
    else
        hipLaunchKernelGGL((spectra_direct<IN_CU8>), grid, block, lds_bytes, st, p);
    return hipGetLastError();
}

}  // namespace rtlws
