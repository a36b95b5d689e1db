// policybench.hip -- the fused kernel's access pattern (per frame: read 2 KiB of
// cmplx_u8, write 4 KiB of f32; one wavefront per frame, 16 wavefronts per CU,
// strided frame order) with no arithmetic, under every cache-policy modifier
// combination of the gfx950 global_load / global_store (sc0, sc1, nt), and with
// the frame read by global_load_lds (LDS-DMA) instead of 2-byte strided loads.
// Build: hipcc -O3 --offload-arch=gfx950 tools/policybench.hip -o tools/build/policybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <chrono>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

#define KERNEL(NAME, LDPOL, STPOL)                                                               \
__global__ __launch_bounds__(64) void NAME(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes) \
{                                                                                                \
    const int t = threadIdx.x;                                                                   \
    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {                                      \
        const uint16_t* src = reinterpret_cast<const uint16_t*>(in) + f * 1024 + t;              \
        unsigned raw[16];                                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                           \
            asm volatile("global_load_ushort %0, %1, off offset:%2 " LDPOL : "=v"(raw[r]) : "v"(src), "n"(128 * r)); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(raw[r]));          \
        float acc[16];                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = (float)(raw[r] & 0xff) + (float)(raw[r] >> 8); \
        f4* dst = reinterpret_cast<f4*>(out + f * 1024) + t;                                     \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                          \
            f4 o = {acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};                 \
            asm volatile("global_store_dwordx4 %0, %1, off offset:%2 " STPOL :: "v"(dst), "v"(o), "n"(1024 * s) : "memory"); \
        }                                                                                        \
    }                                                                                            \
}

KERNEL(k_plain_plain, "", "")
KERNEL(k_nt_nt, "nt", "nt")
KERNEL(k_nt_sc0, "nt", "sc0")
KERNEL(k_nt_sc1, "nt", "sc1")
KERNEL(k_nt_sc0sc1, "nt", "sc0 sc1")
KERNEL(k_nt_sc0nt, "nt", "sc0 nt")
KERNEL(k_nt_sc1nt, "nt", "sc1 nt")
KERNEL(k_nt_sc0sc1nt, "nt", "sc0 sc1 nt")
KERNEL(k_sc0_nt, "sc0", "nt")
KERNEL(k_sc1_nt, "sc1", "nt")
KERNEL(k_sc0sc1_nt, "sc0 sc1", "nt")
KERNEL(k_sc0nt_nt, "sc0 nt", "nt")
KERNEL(k_sc1nt_nt, "sc1 nt", "nt")
KERNEL(k_sc0sc1nt_nt, "sc0 sc1 nt", "nt")
KERNEL(k_plain_nt, "", "nt")

// frame read by two 1 KiB LDS-DMA pieces, lanes then read their 16 points from LDS
template <int AUX>
__global__ __launch_bounds__(64) void k_dma(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    __shared__ __attribute__((aligned(16))) uint16_t stage[1024];
    const int t = threadIdx.x;
    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const uint8_t* src = in + f * 2048;
        __builtin_amdgcn_global_load_lds((glb_vp)(src + t * 16), (lds_vp)stage, 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((glb_vp)(src + 1024 + t * 16), (lds_vp)(stage + 512), 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const unsigned v = stage[64 * r + t]; acc[r] = (float)(v & 0xff) + (float)(v >> 8); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f4* dst = reinterpret_cast<f4*>(out + f * 1024) + t;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 o = {acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
            __builtin_nontemporal_store(o, dst + 64 * s);
        }
    }
}

typedef void (*kern_t)(const uint8_t*, float*, long);

static void run(const char* name, kern_t k, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w : {8, 16}) {
        const int blocks = 256 * w;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, ins[i % 4], outs[i % 4], nframes);
        CHECK(hipDeviceSynchronize());
        const int steps = 200;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, ins[i % 4], outs[i % 4], nframes);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / steps;
        printf("%-34s waves/CU %2d : %7.2f us  %6.0f GB/s\n", name, w, us, 6144.0 * nframes / us / 1e3);
    }
    fflush(stdout);
}

// 16-byte loads: two per lane cover the frame (no transposition: the values end up
// in the wrong lanes, which a timing-only kernel does not mind)
__global__ __launch_bounds__(64) void k_wide(const uint8_t* __restrict__ in, float* __restrict__ out, long nframes)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const u4* src = reinterpret_cast<const u4*>(in + f * 2048);
        const u4 a = __builtin_nontemporal_load(src + t), b = __builtin_nontemporal_load(src + 64 + t);
        float acc[16];
        acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; acc[3] = a.w; acc[4] = b.x; acc[5] = b.y; acc[6] = b.z; acc[7] = b.w;
#pragma unroll
        for (int r = 8; r < 16; ++r) acc[r] = acc[r - 8] * 0.5f;
        f4* dst = reinterpret_cast<f4*>(out + f * 1024) + t;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f4 o = {acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
            __builtin_nontemporal_store(o, dst + 64 * s);
        }
    }
}

// sustained mode for power sampling: policybench <u16|dma|wide|plain|ntplain|sc1|sc0sc1|sc1nt|ldsc1> <seconds>
// (u16 = the product's nt loads + nt stores; ntplain = plain loads + nt stores; sc1.. = store policy; ldsc1 = sc1 loads)
static int sustained(const char* which, double seconds, std::vector<uint8_t*>& ins, std::vector<float*>& outs, long nframes)
{
    kern_t k = !strcmp(which, "dma") ? (kern_t)k_dma<2> : !strcmp(which, "wide") ? (kern_t)k_wide
             : !strcmp(which, "plain") ? (kern_t)k_plain_plain : !strcmp(which, "ntplain") ? (kern_t)k_plain_nt
             : !strcmp(which, "sc1") ? (kern_t)k_nt_sc1 : !strcmp(which, "sc0sc1") ? (kern_t)k_nt_sc0sc1
             : !strcmp(which, "sc1nt") ? (kern_t)k_nt_sc1nt : !strcmp(which, "ldsc1") ? (kern_t)k_sc1_nt
             : (kern_t)k_nt_nt;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0; double ms_total = 0.0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, ins[i % 4], outs[i % 4], nframes);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms; launches += 200;
    }
    printf("%-5s %.2f us per launch over %ld launches\n", which, 1e3 * ms_total / launches, launches);
    return 0;
}

int main(int argc, char** argv)
{
    const long nframes = 65536;
    std::vector<uint8_t*> ins(4); std::vector<float*> outs(4);
    for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&ins[i], nframes * 2048)); CHECK(hipMalloc(&outs[i], nframes * 4096));
        CHECK(hipMemset(ins[i], 0x55 + i, nframes * 2048)); }
    if (argc > 2) return sustained(argv[1], atof(argv[2]), ins, outs, nframes);
    for (int rep = 0; rep < 2; ++rep) {
        run("ld plain      st plain", k_plain_plain, ins, outs, nframes);
        run("ld plain      st nt", k_plain_nt, ins, outs, nframes);
        run("ld nt         st nt   (product)", k_nt_nt, ins, outs, nframes);
        run("ld nt         st sc0", k_nt_sc0, ins, outs, nframes);
        run("ld nt         st sc1", k_nt_sc1, ins, outs, nframes);
        run("ld nt         st sc0 sc1", k_nt_sc0sc1, ins, outs, nframes);
        run("ld nt         st sc0 nt", k_nt_sc0nt, ins, outs, nframes);
        run("ld nt         st sc1 nt", k_nt_sc1nt, ins, outs, nframes);
        run("ld nt         st sc0 sc1 nt", k_nt_sc0sc1nt, ins, outs, nframes);
        run("ld sc0        st nt", k_sc0_nt, ins, outs, nframes);
        run("ld sc1        st nt", k_sc1_nt, ins, outs, nframes);
        run("ld sc0 sc1    st nt", k_sc0sc1_nt, ins, outs, nframes);
        run("ld sc0 nt     st nt", k_sc0nt_nt, ins, outs, nframes);
        run("ld sc1 nt     st nt", k_sc1nt_nt, ins, outs, nframes);
        run("ld sc0 sc1 nt st nt", k_sc0sc1nt_nt, ins, outs, nframes);
        run("ld LDS-DMA aux0  st nt", k_dma<0>, ins, outs, nframes);
        run("ld LDS-DMA nt    st nt", k_dma<2>, ins, outs, nframes);
    }
    return 0;
}
