// box_calib.hip -- bench.py's box calibration (measurement plumbing, NOT part of the product libraries).
//
// Why: the spectrum kernels run at the package power cap, and what the cap buys differs from chip to chip
// (the same kernel reads 0.443-0.487 of the HBM roofline on different boxes).  A fixed,
// memory-free instruction stream characterises the box independently of the product kernels: every SIMD of
// the device holds two wavefronts that issue dependent-free v_fma_f64 on eight register pairs with evolving
// operands (the loop of tools/f64energy.hip, "fma_f64") -- on its own it drives the package to the cap, and
// the shader clock the governor then settles at (d(s_memtime) / d(s_memrealtime) inside the kernel) is the
// box's figure of merit; bench.py reads the package energy accumulator around the same launches for the
// watts.  C-ABI: rtlws_box_calib_run (rtlws_box_calib.h); built into lib/librtlws_bench.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "rtlws_box_calib.h"

namespace {

struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

#define R4(x) x x x x
#define R32(x) R4(R4(x)) R4(R4(x))
#define FMA(r) "v_fma_f64 " #r ", " #r ", %8, %9\n"
#define FMA8 FMA(%0) FMA(%1) FMA(%2) FMA(%3) FMA(%4) FMA(%5) FMA(%6) FMA(%7)

__global__ __launch_bounds__(64, 2) void box_calib_kernel(double* out, Stamp* st, int iters, double a, double b)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    double r0 = 1.0 + 1e-3 * threadIdx.x, r1 = r0 * 1.1, r2 = r0 * 1.2, r3 = r0 * 1.3, r4 = r0 * 1.4, r5 = r0 * 1.5,
           r6 = r0 * 1.6, r7 = r0 * 1.7;
    const double va = a + 1e-9 * threadIdx.x, vb = b - 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {      // 256 v_fma_f64 per iteration: r = r * a + b, values stay O(1)
        asm volatile(R32(FMA8) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                     : "v"(va), "v"(vb));
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (threadIdx.x == 0) {
        st[blockIdx.x].clk0 = c0;
        st[blockIdx.x].clk1 = clock64();
        st[blockIdx.x].rt0 = w0;
        st[blockIdx.x].rt1 = wall_clock64();
    }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" int rtlws_box_calib_run(int device, double seconds, struct rtlws_box_calib* res)
{
    if (!res || !(seconds > 0.0) || seconds > 5.0) return -1;
    *res = rtlws_box_calib{};
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int blocks = 8 * prop.multiProcessorCount;      // two wavefronts per SIMD
    const int iters = 400;                                 // 102 400 instructions per wavefront: ~0.2 ms per launch
    double* out = nullptr;
    Stamp* st = nullptr;
    hipStream_t q = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -3;
    std::vector<Stamp> hst(blocks);
    std::vector<double> ghz;
    long launches = 0;
    float ms = 0.0f;
    if (hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)) != hipSuccess) goto done;
    if (hipMalloc(&st, (size_t)blocks * sizeof(Stamp)) != hipSuccess) goto done;
    if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) goto done;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
    {
        const double t0 = now_s();
        if (hipEventRecord(e0, q) != hipSuccess) goto done;
        // batches of 10 launches (~2 ms) until the time is up: always terminates (bounded by `seconds`)
        while (now_s() - t0 < seconds && launches < 100000) {
            for (int i = 0; i < 10; ++i)
                hipLaunchKernelGGL(box_calib_kernel, dim3(blocks), dim3(64), 0, q, out, st, iters, 0.99999904632568359375, 1.0e-6);
            launches += 10;
            if (hipStreamSynchronize(q) != hipSuccess) goto done;
        }
        if (hipEventRecord(e1, q) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) goto done;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) goto done;
        res->wall_seconds = now_s() - t0;
    }
    // the stamps are the LAST launch's: the clock the governor had settled at
    if (hipMemcpy(hst.data(), st, (size_t)blocks * sizeof(Stamp), hipMemcpyDeviceToHost) != hipSuccess) goto done;
    for (const Stamp& s : hst)
        if (s.rt1 > s.rt0) ghz.push_back((double)(s.clk1 - s.clk0) / ((double)(s.rt1 - s.rt0) * 10.0));   // 100 MHz counter
    if (ghz.empty()) goto done;
    std::sort(ghz.begin(), ghz.end());
    res->sclk_ghz = ghz[ghz.size() / 2];
    res->launches = launches;
    res->gpu_seconds = 1e-3 * ms;
    res->wave_instructions = (double)launches * blocks * iters * 256.0;
    res->simds = 4 * prop.multiProcessorCount;
    rc = 0;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (q) (void)hipStreamDestroy(q);
    if (st) (void)hipFree(st);
    if (out) (void)hipFree(out);
    return rc;
}
