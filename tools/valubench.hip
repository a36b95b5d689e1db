// valubench.hip -- VALU issue cost per instruction class on gfx950, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// 16 independent accumulators v0..v15 style via "+v" operands; each BODY is 16 instrs on distinct regs.
#define DEF_KERNEL(NAME, ASM16)                                                        \
__global__ __launch_bounds__(64) void NAME(float* out, int iters, float a, float b)   \
{                                                                                      \
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    float r8 = r0 + 8, r9 = r0 + 9, r10 = r0 + 10, r11 = r0 + 11, r12 = r0 + 12, r13 = r0 + 13, r14 = r0 + 14, r15 = r0 + 15; \
    for (int i = 0; i < iters; ++i) {                                                  \
        asm volatile(REP16(ASM16)                                                      \
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),       \
              "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11), "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15)  \
            : "v"(a), "v"(b));                                                         \
    }                                                                                  \
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15; \
}

#define I16(op) \
  op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) op(%8) op(%9) op(%10) op(%11) op(%12) op(%13) op(%14) op(%15)

#define ADD_E32(r) "v_add_f32_e32 " #r ", %16, " #r "\n"
#define SUB_E32(r) "v_sub_f32_e32 " #r ", " #r ", %16\n"
#define MUL_LIT(r) "v_mul_f32_e32 " #r ", 0x3f6c835e, " #r "\n"
#define FMAC_E32(r) "v_fmac_f32_e32 " #r ", %16, %17\n"
#define FMAC_LIT(r) "v_fmac_f32_e32 " #r ", 0x3f6c835e, %17\n"
#define FMA_VOP3(r) "v_fma_f32 " #r ", %16, %17, " #r "\n"
#define FMA_NEG(r) "v_fma_f32 " #r ", -%16, %17, " #r "\n"
#define FMAMK(r) "v_fmamk_f32 " #r ", %16, 0x3f3504f3, " #r "\n"
#define CVT_I32(r) "v_cvt_f32_i32_e32 " #r ", " #r "\n"
#define ADD_U32(r) "v_add_u32_e32 " #r ", %16, " #r "\n"
#define ADD_SDWA(r) "v_add_u32_sdwa " #r ", " #r ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define MOV_B32(r) "v_mov_b32_e32 " #r ", %16\n"
#define CVT_UB0(r) "v_cvt_f32_ubyte0_e32 " #r ", " #r "\n"
#define CVT_UB1(r) "v_cvt_f32_ubyte1_e32 " #r ", " #r "\n"
#define PERM(r) "v_perm_b32 " #r ", " #r ", %16, %17\n"
#define ANDOR(r) "v_and_or_b32 " #r ", " #r ", %16, %17\n"
#define MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %16\n"
#define BFE(r) "v_bfe_u32 " #r ", " #r ", 8, 8\n"
#define CNDMASK(r) "v_cndmask_b32_e32 " #r ", %16, " #r ", vcc\n"

DEF_KERNEL(k_add, I16(ADD_E32))
DEF_KERNEL(k_sub, I16(SUB_E32))
DEF_KERNEL(k_mul_lit, I16(MUL_LIT))
DEF_KERNEL(k_fmac, I16(FMAC_E32))
DEF_KERNEL(k_fmac_lit, I16(FMAC_LIT))
DEF_KERNEL(k_fma_vop3, I16(FMA_VOP3))
DEF_KERNEL(k_fma_neg, I16(FMA_NEG))
DEF_KERNEL(k_fmamk, I16(FMAMK))
DEF_KERNEL(k_cvt, I16(CVT_I32))
DEF_KERNEL(k_addu32, I16(ADD_U32))
DEF_KERNEL(k_sdwa, I16(ADD_SDWA))
DEF_KERNEL(k_mov, I16(MOV_B32))


#define DEF_KERNEL64(NAME, ASM8)                                                      \
__global__ __launch_bounds__(64) void NAME(float* out, int iters, float a, float b)   \
{                                                                                      \
    double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    double c = (double)a;                                                              \
    for (int i = 0; i < iters; ++i) {                                                  \
        asm volatile(REP16(ASM8 ASM8)                                                  \
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)        \
            : "v"(c), "v"(a));                                                         \
    }                                                                                  \
    out[blockIdx.x * 64 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7); \
}
#define I8(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7)
#define PKADD(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
#define PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %8\n"
#define MOV64(r) "v_mov_b64_e32 " #r ", %8\n"
#define LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 1, %8\n"
DEF_KERNEL64(k_pkadd, I8(PKADD))
DEF_KERNEL64(k_pkmul, I8(PKMUL))
DEF_KERNEL64(k_pkfma, I8(PKFMA))
DEF_KERNEL64(k_mov64, I8(MOV64))
DEF_KERNEL64(k_lshladd64, I8(LSHLADD64))
#define FMA64(r) "v_fma_f64 " #r ", " #r ", %8, %8\n"
#define ADD64(r) "v_add_f64 " #r ", " #r ", %8\n"
#define MUL64(r) "v_mul_f64 " #r ", " #r ", %8\n"
#define FMAC64(r) "v_fmac_f64_e32 " #r ", %8, %8\n"
#define CVT64U(r) "v_cvt_f64_u32_e32 " #r ", %9\n"
#define CVT64F(r) "v_cvt_f64_f32_e32 " #r ", %9\n"
// dependent chains: the same register 8 times (latency), two registers alternating
#define I8_DEP1(op) op(%0) op(%0) op(%0) op(%0) op(%0) op(%0) op(%0) op(%0)
#define I8_DEP2(op) op(%0) op(%1) op(%0) op(%1) op(%0) op(%1) op(%0) op(%1)
#define I8_DEP4(op) op(%0) op(%1) op(%2) op(%3) op(%0) op(%1) op(%2) op(%3)
DEF_KERNEL64(k_fma64_dep1, I8_DEP1(FMA64))
DEF_KERNEL64(k_fma64_dep2, I8_DEP2(FMA64))
DEF_KERNEL64(k_fma64_dep4, I8_DEP4(FMA64))
DEF_KERNEL64(k_add64_dep1, I8_DEP1(ADD64))
// an f64 stream with one 32-bit integer instruction after every four (the product kernel's mix)
#define MIX5(a, b, c, d) FMA64(a) FMA64(b) FMA64(c) FMA64(d) "v_add_u32_e32 %9, 1, %9\n"
DEF_KERNEL64(k_fma64_mix, MIX5(%0, %1, %2, %3) MIX5(%4, %5, %6, %7))
DEF_KERNEL64(k_fma64, I8(FMA64))
DEF_KERNEL64(k_add64, I8(ADD64))
DEF_KERNEL64(k_mul64, I8(MUL64))
DEF_KERNEL64(k_fmac64, I8(FMAC64))
DEF_KERNEL64(k_cvt64u, I8(CVT64U))
DEF_KERNEL64(k_cvt64f, I8(CVT64F))
DEF_KERNEL(k_cvtub0, I16(CVT_UB0))
DEF_KERNEL(k_cvtub1, I16(CVT_UB1))
DEF_KERNEL(k_perm, I16(PERM))
DEF_KERNEL(k_andor, I16(ANDOR))
DEF_KERNEL(k_mullo, I16(MULLO))
DEF_KERNEL(k_bfe, I16(BFE))
DEF_KERNEL(k_cndmask, I16(CNDMASK))

typedef void (*kern_t)(float*, int, float, float);

void run(const char* name, kern_t k, float* out)
{
    printf("%-14s", name);
    const char* only = getenv("VALUBENCH_ONLY");
    if (only && !strstr(name, only)) return;
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * 4 * wps, iters = 2000;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // instructions per SIMD = wps * iters * 256; report ns per instruction per SIMD
        const double ns_per = 1e6 * ms / ((double)wps * iters * 256);
        printf("  w%d: %6.3f ns/instr", wps, ns_per);
    }
    printf("\n"); fflush(stdout);
}

int main()
{
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float)));
    run("add_e32", k_add, out); run("sub_e32", k_sub, out); run("mul_literal", k_mul_lit, out);
    run("fmac_e32", k_fmac, out); run("fmac_literal", k_fmac_lit, out); run("fma_vop3", k_fma_vop3, out);
    run("fma_vop3_neg", k_fma_neg, out); run("fmamk", k_fmamk, out); run("cvt_f32_i32", k_cvt, out);
    run("pk_add_f32", k_pkadd, out); run("pk_mul_f32", k_pkmul, out); run("pk_fma_f32", k_pkfma, out); run("mov_b64", k_mov64, out); run("lshl_add_u64", k_lshladd64, out);
    run("fma_f64", k_fma64, out); run("fma_f64_dep1", k_fma64_dep1, out); run("fma_f64_dep2", k_fma64_dep2, out); run("fma_f64_dep4", k_fma64_dep4, out);
    run("add_f64_dep1", k_add64_dep1, out); run("fma_f64_mix4+1", k_fma64_mix, out); run("add_f64", k_add64, out); run("mul_f64", k_mul64, out); run("fmac_f64", k_fmac64, out);
    run("cvt_f64_u32", k_cvt64u, out); run("cvt_f64_f32", k_cvt64f, out);
    run("cvt_ubyte0", k_cvtub0, out); run("cvt_ubyte1", k_cvtub1, out); run("perm_b32", k_perm, out); run("and_or_b32", k_andor, out); run("mul_lo_u32", k_mullo, out); run("bfe_u32", k_bfe, out); run("cndmask", k_cndmask, out);
    run("add_u32", k_addu32, out); run("add_u32_sdwa", k_sdwa, out); run("mov_b32", k_mov, out);
    return 0;
}
