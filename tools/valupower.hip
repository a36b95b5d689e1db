// valupower.hip -- does packed f32 math (v_pk_add_f32 / v_pk_fma_f32) cost less
// energy per flop than scalar VALU math?  Runs one instruction mix for a few
// seconds at full occupancy; tools/valupower.sh samples rocm-smi meanwhile.
// usage: valupower <add|pkadd|fma|pkfma> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
typedef float f2v __attribute__((ext_vector_type(2)));

#define REGS16 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
               "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11), "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15)
#define I16(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) op(%8) op(%9) op(%10) op(%11) op(%12) op(%13) op(%14) op(%15)
#define ADD(r) "v_add_f32_e32 " #r ", %16, " #r "\n"
#define FMA(r) "v_fmac_f32_e32 " #r ", %16, %17\n"

template <int MODE>
__global__ __launch_bounds__(64, 4) void k_scalar(float* out, int iters, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float r8 = r0 + 8, r9 = r0 + 9, r10 = r0 + 10, r11 = r0 + 11, r12 = r0 + 12, r13 = r0 + 13, r14 = r0 + 14, r15 = r0 + 15;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16(I16(ADD)) : REGS16 : "v"(a), "v"(b));
        else asm volatile(REP16(I16(FMA)) : REGS16 : "v"(a), "v"(b));
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15;
}

#define P8(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7)
#define PKADD(r) "v_pk_add_f32 " #r ", %8, " #r "\n"
#define PKFMA(r) "v_pk_fma_f32 " #r ", %8, %9, " #r "\n"
#define REGS8 "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)

template <int MODE>
__global__ __launch_bounds__(64, 4) void k_packed(float* out, int iters, float a, float b)
{
    const float t = threadIdx.x;
    f2v q0 = {t, t + 1}, q1 = {t + 2, t + 3}, q2 = {t + 4, t + 5}, q3 = {t + 6, t + 7};
    f2v q4 = {t + 8, t + 9}, q5 = {t + 10, t + 11}, q6 = {t + 12, t + 13}, q7 = {t + 14, t + 15};
    const f2v pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        // 16 bodies of 8 packed instructions = the same 256 lane-operations as the scalar body
        if (MODE == 0) asm volatile(REP16(P8(PKADD)) : REGS8 : "v"(pa), "v"(pb));
        else asm volatile(REP16(P8(PKFMA)) : REGS8 : "v"(pa), "v"(pb));
    }
    const f2v s = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "add";
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    float* out; CHECK(hipMalloc(&out, 4096 * 64 * sizeof(float)));
    const int iters = 2000, blocks = 4096;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (!strcmp(mode, "add")) hipLaunchKernelGGL(k_scalar<0>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.5f, 0.25f);
        else if (!strcmp(mode, "fma")) hipLaunchKernelGGL(k_scalar<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.5f, 0.25f);
        else if (!strcmp(mode, "pkadd")) hipLaunchKernelGGL(k_packed<0>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.5f, 0.25f);
        else hipLaunchKernelGGL(k_packed<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.5f, 0.25f);
    };
    launch(); CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0; double gpu_ms = 0.0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); gpu_ms += ms; launches += 20;
    }
    const double lane_ops = (double)launches * blocks * 64.0 * iters * 256.0;
    printf("%-6s %.3e lane-operations/s  (%ld launches, %.1f ms on the device)\n", mode, lane_ops / (gpu_ms * 1e-3), launches, gpu_ms);
    return 0;
}
