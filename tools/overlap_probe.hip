// overlap_probe.hip -- do f64 VALU work and 16-byte LDS traffic of TWO wavefronts on one SIMD overlap? (diagnostic)
//
// Why: the 4096-point f64 kernels (four wavefronts per frame, two workgroups per CU = two wavefronts per SIMD) run no
// faster with two workgroups per CU than with one (profiles/r06_f64_4096_diag.txt), although neither the vector ALU
// (41 % busy) nor the LDS (20 %) is saturated.  This probe puts two one-wavefront workgroups on every SIMD and gives
// each a role by its wave slot (HW_ID[3:0] & 1): F = a dependent-free v_fma_f64 stream, L = ds_write_b128 /
// ds_read_b128 rounds of a private conflict-free slice (the exchange of spectrum_f64_1024x.hip), M = the kernel's own
// rhythm, 16 writes + 16 reads then 200 FMAs, in one wavefront.  Modes: FF, LL, FL, MM, M- (one wavefront per SIMD).
// Reported: wave64 instructions per second and SIMD of each kind, against the single-role runs.
//
// build: hipcc -O2 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/build/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Rec { unsigned long long clk, rt, fma, lds; unsigned hwid, role; };

#define R4(x) x x x x
#define FMA(r) "v_fma_f64 " #r ", " #r ", %8, %9\n"
#define FMA8 FMA(%0) FMA(%1) FMA(%2) FMA(%3) FMA(%4) FMA(%5) FMA(%6) FMA(%7)

// roles: 0 = F, 1 = L, 2 = M (mixed rhythm), 3 = idle (exits at once)
__global__ __launch_bounds__(64, 2) void probe(Rec* out, int iters, int role_even, int role_odd, double a, double b)
{
    extern __shared__ __attribute__((aligned(16))) double2 lds_all[];
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const int slot = hwid & 15;
    const int role = (slot & 1) ? role_odd : role_even;
    double2* lds = lds_all;            // one workgroup = one wavefront: the whole allocation is its own
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    double r0 = 1.0 + 1e-3 * threadIdx.x, r1 = r0 * 1.1, r2 = r0 * 1.2, r3 = r0 * 1.3, r4 = r0 * 1.4, r5 = r0 * 1.5, r6 = r0 * 1.6, r7 = r0 * 1.7;
    const double va = a + 1e-9 * threadIdx.x, vb = b - 1e-9 * threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = make_double2(a + s + threadIdx.x, b - s);
    const int wp = threadIdx.x >> 4, wc = threadIdx.x & 15;
    unsigned long long nf = 0, nl = 0;
    if (role == 0) {
        for (int i = 0; i < iters; ++i) {
            asm volatile(R4(R4(FMA8)) R4(R4(FMA8)) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(va), "v"(vb));
            nf += 256;
        }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
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
            nl += 128;
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) lds[17 * (4 * s + wp) + wc] = v[s];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = lds[17 * threadIdx.x + c];
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(v[c].x), "+v"(v[c].y));
            asm volatile(R4(R4(FMA8)) R4(FMA8) R4(FMA8) FMA8 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(va), "v"(vb));
            nf += 200;
            nl += 32;
        }
    }
    double s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += v[c].x + v[c].y;
    if (s == 12345.6789) out[0].clk = 0;
    if (threadIdx.x == 0) {
        Rec r;
        r.clk = clock64() - c0; r.rt = wall_clock64() - w0; r.fma = nf; r.lds = nl; r.hwid = hwid; r.role = role;
        out[blockIdx.x] = r;
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = 8 * p.multiProcessorCount;
    const size_t lds = 16 * 17 * 64;
    Rec* d; CHECK(hipMalloc(&d, blocks * sizeof(Rec)));
    std::vector<Rec> h(blocks);
    struct Mode { const char* name; int even, odd; } modes[] = {
        {"FF  fma | fma", 0, 0}, {"LL  lds | lds", 1, 1}, {"FL  fma | lds", 0, 1}, {"MM  mix | mix", 2, 2},
        {"F-  fma | idle", 0, 3}, {"L-  lds | idle", 1, 3}, {"M-  mix | idle", 2, 3}};
    printf("%-16s %12s %12s %10s %10s   (per SIMD: wave64 instructions per shader clock; clocks per instruction)\n", "mode", "fma/clk/SIMD", "lds/clk/SIMD", "clk/fma", "clk/lds");
    for (const Mode& m : modes) {
        for (int rep = 0; rep < 3; ++rep) {       // warm, then measure the last
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), lds, 0, d, 2000, m.even, m.odd, 0.99999904632568359375, 1.0e-6);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
        // per SIMD: sum the instructions of its wavefronts, divide by the longest wavefront's clocks
        double fma = 0, ldsn = 0, clk = 0; int roles[4] = {0, 0, 0, 0};
        double rclk[4] = {0, 0, 0, 0};
        for (const Rec& r : h) { fma += (double)r.fma; ldsn += (double)r.lds; if ((double)r.clk > clk) clk = (double)r.clk; roles[r.role & 3]++; rclk[r.role & 3] += (double)r.clk; }
        const double simds = 4.0 * p.multiProcessorCount;
        printf("    mean wavefront clocks by role: F %.0f  L %.0f  M %.0f\n", roles[0] ? rclk[0] / roles[0] : 0.0, roles[1] ? rclk[1] / roles[1] : 0.0, roles[2] ? rclk[2] / roles[2] : 0.0);
        printf("%-16s %12.4f %12.4f %10.2f %10.2f   roles F/L/M/idle %d/%d/%d/%d, longest wavefront %.0f clocks\n", m.name, fma / simds / clk, ldsn / simds / clk,
               fma > 0 ? simds * clk / fma : 0.0, ldsn > 0 ? simds * clk / ldsn : 0.0, roles[0], roles[1], roles[2], roles[3], clk);
    }
    return 0;
}
