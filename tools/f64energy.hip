// f64energy.hip -- energy per wave64 instruction, by class, on gfx950 (diagnostic).
//
// Why: spectrum_f64_1024x.hip runs at the package power limit with the clock below its knee
// (profiles/r05_ab_waves_lds_counter.txt: 7 % fewer shader cycles per launch came back as a 7 % lower
// clock and the same time), so what a launch costs is ENERGY, and which instruction mix is the
// cheapest cannot be read from issue cycles.  Every mode runs the same shape -- 2 048 one-wavefront
// workgroups (2 per SIMD), a loop of 256 instructions of one class on sixteen (eight 64-bit) independent
// registers with evolving operands -- for a few seconds while the host reads the package energy
// accumulator (rocm_smi: rsmi_dev_energy_count_get) and the socket power; the in-kernel clock is
// d(s_memtime) / d(s_memrealtime).  Reported: instructions per second and SIMD, watts, nanojoules per
// wave64 instruction, and the same with the "nop" mode's power at that clock subtracted.
//
// build: hipcc -O2 --offload-arch=gfx950 tools/f64energy.hip -o tools/build/f64energy -lrocm_smi64
// usage: f64energy [seconds per mode] [mode ...]
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)

struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

#define PROLOGUE                                                                                         \
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#define EPILOGUE                                                                                         \
    if (threadIdx.x == 0) { st[blockIdx.x].clk0 = c0; st[blockIdx.x].clk1 = clock64();                   \
                            st[blockIdx.x].rt0 = w0; st[blockIdx.x].rt1 = wall_clock64(); }

// ---- 64-bit classes: eight independent double registers, operands a (VGPR pair), s (SGPR pair)
#define D8(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7)
#define DEF64(NAME, BODY8)                                                                               \
__global__ __launch_bounds__(64, 2) void NAME(double* out, Stamp* st, int iters, double a, double b)    \
{                                                                                                        \
    PROLOGUE                                                                                             \
    double r0 = 1.0 + 1e-3 * threadIdx.x, r1 = r0 * 1.1, r2 = r0 * 1.2, r3 = r0 * 1.3, r4 = r0 * 1.4, r5 = r0 * 1.5, r6 = r0 * 1.6, r7 = r0 * 1.7; \
    double va = a + 1e-9 * threadIdx.x, vb = b - 1e-9 * threadIdx.x;                                    \
    int vi = threadIdx.x;                                                                                \
    for (int i = 0; i < iters; ++i) {                                                                    \
        asm volatile(REP32(BODY8)                                                                        \
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(vi)  \
            : "v"(va), "v"(vb), "s"(a), "s"(b));                                                         \
    }                                                                                                    \
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + vi;                    \
    EPILOGUE                                                                                             \
}
// r = r * a + b with a = 1 - 2^-20-ish, b small: values stay O(1) and every mantissa bit moves
#define ADD64(r)   "v_add_f64 " #r ", " #r ", %10\n"
#define ADD64S(r)  "v_add_f64 " #r ", " #r ", %12\n"
#define MUL64(r)   "v_mul_f64 " #r ", " #r ", %9\n"
#define MUL64S(r)  "v_mul_f64 " #r ", " #r ", %11\n"
#define FMA64(r)   "v_fma_f64 " #r ", " #r ", %9, %10\n"
#define FMA64S(r)  "v_fma_f64 " #r ", " #r ", %11, %10\n"
#define FMA64L(r)  "v_fma_f64 " #r ", " #r ", 0.5, %10\n"
#define FMAC64(r)  "v_fmac_f64_e32 " #r ", %9, %10\n"
#define CVT64(r)   "v_cvt_f64_i32_e32 " #r ", %8\n"
#define MOV64(r)   "v_mov_b64_e32 " #r ", %9\n"
DEF64(k_add64, D8(ADD64))
DEF64(k_add64_s, D8(ADD64S))
DEF64(k_mul64, D8(MUL64))
DEF64(k_mul64_s, D8(MUL64S))
DEF64(k_fma64, D8(FMA64))
DEF64(k_fma64_s, D8(FMA64S))
DEF64(k_fma64_lit, D8(FMA64L))
DEF64(k_fmac64, D8(FMAC64))
DEF64(k_cvt64, D8(CVT64))
DEF64(k_mov64, D8(MOV64))
// the product kernel's rough mix: 5 fma : 2 add : 1 mul
#define MIX8 FMA64(%0) ADD64(%1) FMA64(%2) FMA64(%3) MUL64(%4) FMA64(%5) ADD64(%6) FMA64(%7)
DEF64(k_mix64, MIX8)
// wave idle in the loop: s_nop only (what the clock tree, the sequencer and leakage cost at that clock)
#define NOP8 "s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n"
DEF64(k_nop, NOP8)

// ---- 32-bit classes: sixteen independent registers
#define I16(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) op(%8) op(%9) op(%10) op(%11) op(%12) op(%13) op(%14) op(%15)
#define DEF32(NAME, BODY16)                                                                              \
__global__ __launch_bounds__(64, 2) void NAME(double* out, Stamp* st, int iters, double a, double b)    \
{                                                                                                        \
    PROLOGUE                                                                                             \
    float r0 = 1.0f + 1e-3f * threadIdx.x, r1 = r0 * 1.1f, r2 = r0 * 1.2f, r3 = r0 * 1.3f, r4 = r0 * 1.4f, r5 = r0 * 1.5f, r6 = r0 * 1.6f, r7 = r0 * 1.7f; \
    float r8 = r0 * 1.8f, r9 = r0 * 1.9f, r10 = r0 * 2.0f, r11 = r0 * 2.1f, r12 = r0 * 2.2f, r13 = r0 * 2.3f, r14 = r0 * 2.4f, r15 = r0 * 2.5f; \
    float fa = (float)a + 1e-6f * threadIdx.x, fb = (float)b;                                            \
    for (int i = 0; i < iters; ++i) {                                                                    \
        asm volatile(REP16(BODY16)                                                                       \
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),           \
              "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11), "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15)      \
            : "v"(fa), "v"(fb));                                                                         \
    }                                                                                                    \
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15; \
    EPILOGUE                                                                                             \
}
#define FMA32(r)   "v_fma_f32 " #r ", " #r ", %16, %17\n"
#define ADD32(r)   "v_add_f32_e32 " #r ", %17, " #r "\n"
#define PKADD16(r) "v_pk_add_u16 " #r ", " #r ", %16\n"
#define PERM(r)    "v_perm_b32 " #r ", " #r ", %16, %17\n"
#define MOV32(r)   "v_mov_b32_e32 " #r ", %16\n"
#define ADDU32(r)  "v_add_u32_e32 " #r ", %16, " #r "\n"
DEF32(k_fma32, I16(FMA32))
DEF32(k_add32, I16(ADD32))
DEF32(k_pkadd16, I16(PKADD16))
DEF32(k_perm, I16(PERM))
DEF32(k_mov32, I16(MOV32))
DEF32(k_addu32, I16(ADDU32))
#define SWAP32_8 "v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_permlane32_swap_b32_e32 %4, %5\n v_permlane32_swap_b32_e32 %6, %7\n" \
                 "v_permlane16_swap_b32_e32 %8, %9\n v_permlane16_swap_b32_e32 %10, %11\n v_permlane16_swap_b32_e32 %12, %13\n v_permlane16_swap_b32_e32 %14, %15\n"
DEF32(k_swap, SWAP32_8 SWAP32_8)

// ---- LDS: 16 x (ds_write_b128 + ds_read_b128) of a wavefront-private, conflict-free slice (rows of 17)
__global__ __launch_bounds__(64, 2) void k_lds128(double* out, Stamp* st, int iters, double a, double b)
{
    PROLOGUE
    extern __shared__ __attribute__((aligned(16))) double2 lds[];
    double2 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = make_double2(a + s + threadIdx.x, b - s);
    const int wp = threadIdx.x >> 4, wc = threadIdx.x & 15;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) lds[17 * (4 * s + wp) + wc] = v[s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = lds[17 * threadIdx.x + c];
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += v[c].x + v[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    EPILOGUE
}

typedef void (*kern_t)(double*, Stamp*, int, double, double);
struct Mode { const char* name; kern_t k; int per_iter; size_t lds; };     // per_iter: wave64 instructions of the class per loop iteration

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const Mode modes[] = {
        {"nop", k_nop, 256, 0}, {"add_f64", k_add64, 256, 0}, {"add_f64_sgpr", k_add64_s, 256, 0},
        {"mul_f64", k_mul64, 256, 0}, {"mul_f64_sgpr", k_mul64_s, 256, 0}, {"fma_f64", k_fma64, 256, 0},
        {"fma_f64_sgpr", k_fma64_s, 256, 0}, {"fma_f64_inline", k_fma64_lit, 256, 0}, {"fmac_f64", k_fmac64, 256, 0},
        {"cvt_f64_i32", k_cvt64, 256, 0}, {"mov_b64", k_mov64, 256, 0}, {"mix_5fma_2add_1mul", k_mix64, 256, 0},
        {"fma_f32", k_fma32, 256, 0}, {"add_f32", k_add32, 256, 0}, {"pk_add_u16", k_pkadd16, 256, 0},
        {"perm_b32", k_perm, 256, 0}, {"mov_b32", k_mov32, 256, 0}, {"add_u32", k_addu32, 256, 0},
        {"permlane_swap", k_swap, 256, 0}, {"lds_w128_r128", k_lds128, 256, 16 * 17 * 64},
    };
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("rsmi_init failed\n"); return 2; }
    const int blocks = 2048, iters = 400;
    double* out; Stamp* st;
    CHECK(hipMalloc(&out, blocks * 64 * sizeof(double)));
    CHECK(hipMalloc(&st, blocks * sizeof(Stamp)));
    std::vector<Stamp> hst(blocks);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-20s %10s %8s %8s %9s %10s %12s\n", "mode", "Ginstr/s", "sclk", "watts", "W(smi)", "nJ/instr", "cyc/instr/SIMD");
    for (const Mode& m : modes) {
        bool want = argc <= 2;
        for (int i = 2; i < argc; ++i) want = want || !strcmp(argv[i], m.name);
        if (!want) continue;
        auto launch = [&]() { hipLaunchKernelGGL(m.k, dim3(blocks), dim3(64), m.lds, 0, out, st, iters, 0.99999904632568359375, 1.0e-6); };
        // settle the governor
        const double tw = now_s();
        while (now_s() - tw < 1.0) { for (int i = 0; i < 20; ++i) launch(); CHECK(hipDeviceSynchronize()); }
        uint64_t c0 = 0, c1 = 0, ts0 = 0, ts1 = 0; float res = 0;
        const bool have_energy = rsmi_dev_energy_count_get(0, &c0, &res, &ts0) == RSMI_STATUS_SUCCESS;
        const double t0 = now_s();
        long launches = 0; double gpu_ms = 0.0, wsum = 0.0; int wn = 0;
        while (now_s() - t0 < seconds) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); gpu_ms += ms; launches += 50;
            uint64_t pw = 0;
            if (rsmi_dev_current_socket_power_get(0, &pw) == RSMI_STATUS_SUCCESS) { wsum += pw * 1e-6; ++wn; }
        }
        const double t1 = now_s();
        double joules = 0;
        if (have_energy && rsmi_dev_energy_count_get(0, &c1, &res, &ts1) == RSMI_STATUS_SUCCESS) joules = (double)(c1 - c0) * res * 1e-6;
        CHECK(hipMemcpy(hst.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> ghz;
        for (const Stamp& s : hst) if (s.rt1 > s.rt0) ghz.push_back((double)(s.clk1 - s.clk0) / ((double)(s.rt1 - s.rt0) * 10.0));
        std::sort(ghz.begin(), ghz.end());
        const double sclk = ghz.empty() ? 0 : ghz[ghz.size() / 2];
        const double instr = (double)launches * blocks * iters * m.per_iter;       // wave64 instructions
        const double watts = joules / (t1 - t0);
        // the device was busy gpu_ms of the wall interval; energy per instruction from the busy share
        const double busy = gpu_ms * 1e-3 / (t1 - t0);
        printf("%-20s %10.2f %8.3f %8.0f %9.0f %10.3f %12.2f   (busy %.3f)\n", m.name, instr / (gpu_ms * 1e-3) * 1e-9, sclk, watts,
               wn ? wsum / wn : 0.0, joules / instr * 1e9, sclk * 1e9 * (gpu_ms * 1e-3) * 1024.0 / instr, busy);
        fflush(stdout);
    }
    return 0;
}
