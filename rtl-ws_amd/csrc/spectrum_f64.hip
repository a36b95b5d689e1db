// spectrum_f64.hip -- the f64 ("exact") power-spectrum kernel, any 2 <= N <= 8192.
//
// The reference computes in double throughout: conversion to double
// (reference src/spectrum.c:54-58,72-76,90-94), an f64 forward DFT (FFTW,
// :21), |X|^2 accumulated into the caller's double buffer (:23-34), and the
// dB / truncate / clamp epilogue in double (src/cbb_main.c:112,121-130).  The
// reference-API paths (spectrum_add_*, cbb_main.h) move one to six frames per
// call, so their arithmetic rate is irrelevant and they run here, in the
// reference's own precision; the f32 fused kernel (spectrum_fused.hip) stays
// the throughput path of the batch API.
//
// One 256-thread workgroup per output row.  A frame lives in LDS as N complex
// doubles.  N a power of two: in-place radix-2 decimation-in-time (bit-reversed
// load, log2 N butterfly stages, twiddles W_N^k from an f64 table built on the
// host in long double).  Any other N: the O(N^2) sum with the same table and an
// exact index walk (e += k mod N), like spectrum_direct.hip but in double.
// K-frame accumulation and the DC-slot rule (src/spectrum.c:25-33 in closed
// form: slot N/2 = sum_k (K-k) P_k[N-1]) are kept in registers, one value per
// output slot a thread owns (at most 32).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "rtlws_internal.h"

namespace rtlws {

typedef double2 d2;

__device__ __forceinline__ d2 mkd(double x, double y) { return make_double2(x, y); }

// sample n of a frame as the reference converts it (before the 1/128 scale)
template <int IN>
__device__ __forceinline__ d2 load_sample(const SpectraParamsF64& p, long frame, int n)
{
    const int N = p.n_fft;
    if constexpr (IN == IN_CS32) {
        const int2 s = reinterpret_cast<const int2*>(p.in)[frame * N + n];
        return mkd((double)s.x, (double)s.y);
    } else if constexpr (IN == IN_RF32) {
        return mkd((double)reinterpret_cast<const float*>(p.in)[frame * N + n], 0.0);
    } else {
        const int R = p.cic_r;
        const uint8_t* q = reinterpret_cast<const uint8_t*>(p.in) + ((frame * N + n) * (long)R) * 2;
        int si = 0, sq = 0;
        for (int r = 0; r < R; ++r) { si += (int)q[2 * r] - 128; sq += (int)q[2 * r + 1] - 128; }
        return mkd((double)si, (double)sq);
    }
}

template <int IN>
__global__ __launch_bounds__(256) void spectra_f64(const SpectraParamsF64 p)
{
    extern __shared__ __attribute__((aligned(16))) double2 xs[];
    constexpr int MAXJ = 32;                  // slots per thread: N <= 8192
    const int N = p.n_fft;
    const int K = p.k_avg;
    const int LOG2N = p.log2n;                // 0: N is not a power of two
    const int tid = threadIdx.x;
    const long g = blockIdx.x;

    double acc[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) acc[j] = 0.0;

    for (int kf = 0; kf < K; ++kf) {
        const long frame = g * K + kf;
        __syncthreads();                      // previous frame's readers are done
        for (int n = tid; n < N; n += 256) {
            d2 v = load_sample<IN>(p, frame, n);
            v = mkd(v.x * p.in_scale, v.y * p.in_scale);     // /128 is exact (power of two)
            if (p.window) { const double w = p.window[n]; v = mkd(v.x * w, v.y * w); }
            int dst = n;
            if (LOG2N) dst = (int)(__brev((unsigned)n) >> (32 - LOG2N));
            xs[dst] = v;
        }
        __syncthreads();

        if (LOG2N) {
            for (int s = 1; s <= LOG2N; ++s) {
                const int half = 1 << (s - 1);
                const int tstep = N >> s;     // W_(2*half)^pos = W_N^(pos * N / (2*half))
                for (int b = tid; b < N / 2; b += 256) {
                    const int pos = b & (half - 1);
                    const int i0 = ((b >> (s - 1)) << s) + pos;
                    const int i1 = i0 + half;
                    const d2 w = p.tw[pos * tstep];
                    const d2 u = xs[i0], x1 = xs[i1];
                    const d2 t = mkd(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
                    xs[i0] = mkd(u.x + t.x, u.y + t.y);
                    xs[i1] = mkd(u.x - t.x, u.y - t.y);
                }
                __syncthreads();
            }
        }

        // |X|^2 per owned slot; slot i shows bin (i + N/2) % N (src/spectrum.c:25),
        // the slot of bin 0 mirrors bin N-1 with the running-sum weights (K - kf).
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int i = tid + 256 * j;
            if (i < N) {
                const int k = (i + N / 2) % N;
                const int kk = (k == 0) ? N - 1 : k;
                d2 X;
                if (LOG2N) {
                    X = xs[kk];
                } else {
                    double ar = 0.0, ai = 0.0;
                    int e = 0;
                    for (int n = 0; n < N; ++n) {
                        const d2 w = p.tw[e];
                        const d2 x = xs[n];
                        ar += x.x * w.x - x.y * w.y;
                        ai += x.x * w.y + x.y * w.x;
                        e += kk;
                        if (e >= N) e -= N;
                    }
                    X = mkd(ar, ai);
                }
                const double pw = X.x * X.x + X.y * X.y;
                acc[j] += (k == 0) ? (double)(K - kf) * pw : pw;
            }
        }
    }

#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int i = tid + 256 * j;
        if (i < N) {
            const double a = acc[j];
            if (p.out_mode == OUT_PAYLOAD) {
                // src/cbb_main.c:125-128, same operation order, in double
                const double d = 10.0 * log10(fabs(p.lin_gain * a / (double)p.count));
                const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
                reinterpret_cast<uint8_t*>(p.out)[g * N + i] = (uint8_t)m;
            } else {
                const double o = (p.out_mode == OUT_DB) ? 10.0 * log10(a / (double)p.count) : a;
                if (p.rows_f32) reinterpret_cast<float*>(p.out)[g * N + i] = (float)o;   // RTLWS_FLAG_ROWS_F32
                else reinterpret_cast<double*>(p.out)[g * N + i] = o;
            }
        }
    }
}

// dB / truncate / clamp of reference src/cbb_main.c:125-128 on f64 sums.
__global__ __launch_bounds__(256) void payload_f64_kernel(const double* __restrict__ sums, int n,
                                                          double gain, int count,
                                                          uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double d = 10.0 * log10(fabs(gain * sums[i] / (double)count));
        const unsigned m = (d >= 0.0) ? (d <= 255.0 ? (unsigned)(int)d : 255u) : 0u;
        out[i] = (uint8_t)m;
    }
}

hipError_t launch_payload_f64(const double* d_sums, int n, double gain, int count, uint8_t* d_out,
                              hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(payload_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_sums, n, gain,
                       count, d_out);
    return hipGetLastError();
}

// Welch accumulation across launches (cbb_main.h, RTLWS_CBB_ALL_FRAMES=2): every
// sensor buffer of a 250 ms interval contributes one row of K_j-frame sums; the
// published row must be what the reference's sequential loop (src/spectrum.c:25-33,
// src/cbb_main.c:50-59) would leave after ALL those frames in one zeroed buffer.
// Every slot but N/2 is a plain sum.  Slot N/2 is sum_k (Ktot - k) P_k[N-1]; with
// launch j covering frames s_j .. s_j+K_j-1 its own output there is
// dc_j = sum_i (K_j - i) P, so the total is
//   sum_j dc_j + Ktot * sum_j S_j - sum_j (s_j + K_j) S_j,   S_j = row_j[N/2 - 1],
// i.e. the accumulated slot plus Ktot times the accumulated neighbour minus a
// scalar B that is carried beside the row.
__global__ __launch_bounds__(256) void welch_accumulate_kernel(double* __restrict__ acc,
                                                               const double* __restrict__ part, int n,
                                                               double frames_end, double* __restrict__ b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += part[i];
    if (i == 0) *b += part[n / 2 - 1] * frames_end;
}

__global__ void welch_finish_kernel(double* __restrict__ acc, int n, double total, double* __restrict__ b)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc[n / 2] += total * acc[n / 2 - 1] - *b;
        *b = 0.0;
    }
}

hipError_t launch_welch_accumulate(double* d_acc, const double* d_part, int n, long frames_end,
                                   double* d_b, hipStream_t st)
{
    hipLaunchKernelGGL(welch_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_acc, d_part, n,
                       (double)frames_end, d_b);
    return hipGetLastError();
}

hipError_t launch_welch_finish(double* d_acc, int n, long total, double* d_b, hipStream_t st)
{
    hipLaunchKernelGGL(welch_finish_kernel, dim3(1), dim3(64), 0, st, d_acc, n, (double)total, d_b);
    return hipGetLastError();
}

template <int IN>
static hipError_t launch_f64_in(const SpectraParamsF64& p, hipStream_t st, int device)
{
    const size_t lds_bytes = sizeof(double2) * (size_t)p.n_fft;
    if (lds_bytes > 64 * 1024) {       // 4096 < N <= 8192: raised once per device to the most any N needs
        static std::atomic<unsigned long long> ready{0};
        const unsigned long long bit = 1ull << (device & 63);
        if (!(ready.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectra_f64<IN>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double2) * 8192));
            if (e != hipSuccess) return e;
            ready.fetch_or(bit, std::memory_order_release);
        }
    }
    if (p.ngroups <= 0) return hipSuccess;   // rtlws_engine_prepare_f64: the attribute only, nothing enqueued
    hipLaunchKernelGGL((spectra_f64<IN>), dim3((unsigned)p.ngroups), dim3(256), lds_bytes, st, p);
    return hipGetLastError();
}

hipError_t launch_spectra_f64(const SpectraParamsF64& p, int in_kind, hipStream_t st, int device)
{
    if (in_kind == IN_CS32) return launch_f64_in<IN_CS32>(p, st, device);
    if (in_kind == IN_RF32) return launch_f64_in<IN_RF32>(p, st, device);
    return launch_f64_in<IN_CU8>(p, st, device);
}

}  // namespace rtlws
