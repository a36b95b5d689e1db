// shim.hip -- the extern-"C" layer declared in include/rtlws_hip.h.
//
// Host C code (rtl-ws_amd/host/*.c) and the Python test/bench plumbing reach
// the kernels only through these functions.  The twiddle tables are computed
// here in f64 and rounded once to f32.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rtlws_hip.h"
#include "rtlws_internal.h"

namespace {

thread_local std::string g_err;

void set_err(const char* what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_err = buf;
}

#define HIP_TRY(expr, ret)                      \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) {                 \
            set_err(#expr, _e);                 \
            return ret;                         \
        }                                       \
    } while (0)

#define NEED_ENGINE(e, ret)                                   \
    do {                                                      \
        if (!(e)) {                                           \
            g_err = std::string(__func__) + ": null engine";  \
            return ret;                                       \
        }                                                     \
    } while (0)

struct Tables {
    float2* tw1 = nullptr;          // fused: input scale 1 (real f32); direct: W_N^e
    float2* tw1_128 = nullptr;      // fused: input scale 1/128 (u8, s32, CIC)
    float2* tw2 = nullptr;          // fused: last-pass (c, s/c) pairs
    float* hann = nullptr;
    float2* hann_cs = nullptr;      // fused: [T] (0.5 cos, 0.5 sin)(2 pi t / N)
    double2* tw64 = nullptr;        // f64 kernel: W_N^k, k < N
    double* hann64 = nullptr;
    double2* tw1_64 = nullptr;      // f64 fused kernel: [T][16] W_N^(t*rev16(s)) / 128
    double2* tw1u_64 = nullptr;     //   ... unscaled (real f32 input)
    double2* tw2_64 = nullptr;      //   [16][R3/2] last-pass (cos, sin/cos) pairs
    double2* hann_cs64 = nullptr;   //   [T] (0.5 cos, 0.5 sin)(2 pi t / N)
    double2* twxa_64 = nullptr;     // spectrum_f64_1024x.hip (N = 1024): [4][8] pass-A (cos, tan) pairs
    double2* twxb_64 = nullptr;     //   [64][16] inner twiddles x lane constant / 128
};

constexpr double kTwoPi = 6.283185307179586476925286766559;

}  // namespace

// Kernel-selection switches (experiments, A/B runs, tests).  Read from the environment ONCE, when
// the engine is created, and changed afterwards only through rtlws_engine_set_option: nothing on a
// launch path calls getenv (tests/test_abi_cpu.py checks the library's imports per function).
struct EngineOpts {
    int v2 = -1;                 // RTLWS_V2: -1 = default rule (K = 1 rows), 0 / 1 = never / always
    int blocks_per_cu = 0;       // RTLWS_BLOCKS_PER_CU: > 0 overrides the f32 fused kernels' grid
    int f64_fused = 1;           // RTLWS_F64_FUSED=0: f64 batches stay on the row-per-workgroup kernel
    int f64_blocks_per_cu = 0;   // RTLWS_F64_BLOCKS_PER_CU
    int f64_x1024 = 1;           // RTLWS_F64_X1024=0: rectangular 1024-point u8 frames stay on the two-transposition kernel
    int f64_x_waves = 0;         // RTLWS_F64_X_WAVES: wavefronts per workgroup of that kernel: 0 = by batch size, 1, 8
    bool f64_x_waves8_ok = true; // the device's LDS limit per workgroup holds the eight-wavefront form (136 KiB)
    int cic_direct = 0;          // RTLWS_CIC_DIRECT=1: every R != 8 on per-lane direct loads
    int cic_round = 0;           // RTLWS_CIC_ROUND=1|2|4: LDS staging depth where R fits it
};

struct rtlws_engine {
    int device = 0;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    EngineOpts opt;
    std::mutex mu;
    std::map<int, Tables> tables;   // by n_fft (fused) or -n_fft (direct)
};

namespace {

int rev16h(int s) { return 4 * (s & 3) + (s >> 2); }

bool is_fused_n(int n) { return n == 1024 || n == 2048 || n == 4096; }

// (cos, sin/cos) of -2*pi*num/den, the form the last pass multiplies by
// (spectrum_fused.hip, "last pass"); cos = 0 is stored as 1e-20.
float2 cos_tan_pair(long num, long den)
{
    num %= den;
    double c, sn;
    if (4 * num == den) { c = 0.0; sn = -1.0; }
    else if (4 * num == 3 * den) { c = 0.0; sn = 1.0; }
    else if (2 * num == den) { c = -1.0; sn = 0.0; }
    else if (num == 0) { c = 1.0; sn = 0.0; }
    else {
        const double a = -kTwoPi * (double)num / (double)den;
        c = std::cos(a);
        sn = std::sin(a);
    }
    if (c == 0.0) c = 1e-20;
    return make_float2((float)c, (float)(sn / c));
}

template <typename V>
bool upload_table(const std::vector<V>& host, V** dev)
{
    hipError_t err = hipMalloc(dev, host.size() * sizeof(V));
    if (err == hipSuccess) err = hipMemcpy(*dev, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice);
    if (err != hipSuccess) set_err("twiddle table upload", err);
    return err == hipSuccess;
}

void free_tables(Tables& tb)
{
    (void)hipFree(tb.tw1);
    (void)hipFree(tb.tw1_128);
    (void)hipFree(tb.tw2);
    (void)hipFree(tb.hann);
    (void)hipFree(tb.hann_cs);
    (void)hipFree(tb.tw64);
    (void)hipFree(tb.hann64);
    (void)hipFree(tb.tw1_64);
    (void)hipFree(tb.tw1u_64);
    (void)hipFree(tb.tw2_64);
    (void)hipFree(tb.hann_cs64);
    (void)hipFree(tb.twxa_64);
    (void)hipFree(tb.twxb_64);
    tb = Tables();
}

// Build (once per engine and N) the per-thread twiddle tables of the fused
// kernel: tw1[t][s] = scale * W_N^(t * rev16(s)) (s >= 1; the kernel scales slot
// 0 itself), tw2[q2][.] = the R3/2 pairs of alpha = W_T^q2: for sub-size
// L = 2, 4, .. R3 and p < max(1, L/4) the pair of alpha^(R3/L) * W_L^p.
int get_tables(rtlws_engine* e, int n_fft, bool fused, Tables* out)
{
    std::lock_guard<std::mutex> lk(e->mu);
    const int key = fused ? n_fft : -n_fft;
    auto it = e->tables.find(key);
    if (it != e->tables.end()) { *out = it->second; return 0; }
    Tables tb;
    HIP_TRY(hipSetDevice(e->device), -3);
    if (fused) {
        const int T = n_fft / 16, R3 = n_fft / 256, NP = R3 / 2;
        std::vector<float2> h1((size_t)T * 16), h1s((size_t)T * 16), h2((size_t)NP * 16);
        for (int t = 0; t < T; ++t)
            for (int s = 0; s < 16; ++s) {
                const long ex = ((long)t * rev16h(s)) % n_fft;
                const double a = -kTwoPi * (double)ex / (double)n_fft;
                const float c = (float)std::cos(a), sn = (float)std::sin(a);
                h1[(size_t)t * 16 + s] = make_float2(c, sn);
                h1s[(size_t)t * 16 + s] = make_float2(c * 0.0078125f, sn * 0.0078125f);
            }
        for (int q2 = 0; q2 < 16; ++q2) {
            int k = 0;
            for (int L = 2; L <= R3; L *= 2)
                for (int p = 0; p < (L >= 4 ? L / 4 : 1); ++p)
                    // alpha^(R3/L) * W_L^p = W_(T*L)^(q2*R3 + p*T)
                    h2[(size_t)q2 * NP + k++] = cos_tan_pair((long)q2 * R3 + (long)p * T, (long)T * L);
        }
        std::vector<float2> hcs((size_t)T);
        for (int t = 0; t < T; ++t) {
            const double a = kTwoPi * (double)t / (double)n_fft;
            hcs[t] = make_float2((float)(0.5 * std::cos(a)), (float)(0.5 * std::sin(a)));
        }
        if (!upload_table(h1, &tb.tw1) || !upload_table(h1s, &tb.tw1_128) || !upload_table(h2, &tb.tw2) ||
            !upload_table(hcs, &tb.hann_cs)) {
            free_tables(tb);
            return -3;
        }
    } else {
        std::vector<float2> h1((size_t)n_fft);
        for (int k = 0; k < n_fft; ++k) {
            const double a = -kTwoPi * (double)k / (double)n_fft;
            h1[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        if (!upload_table(h1, &tb.tw1)) return -3;
    }
    {
        std::vector<float> hw((size_t)n_fft);
        for (int n = 0; n < n_fft; ++n)
            hw[n] = (float)(0.5 - 0.5 * std::cos(kTwoPi * (double)n / (double)n_fft));
        if (!upload_table(hw, &tb.hann)) {
            free_tables(tb);
            return -3;
        }
    }
    e->tables[key] = tb;
    *out = tb;
    return 0;
}

// Tables of the f64 kernel (spectrum_f64.hip), once per engine and N: W_N^k for
// k < N evaluated in long double and rounded once (axis values exact); periodic
// Hann in double.
constexpr int kF64Key = 1 << 24;

int get_tables_f64(rtlws_engine* e, int n_fft, Tables* out)
{
    std::lock_guard<std::mutex> lk(e->mu);
    auto it = e->tables.find(kF64Key + n_fft);
    if (it != e->tables.end()) { *out = it->second; return 0; }
    Tables tb;
    HIP_TRY(hipSetDevice(e->device), -3);
    std::vector<double2> hw((size_t)n_fft);
    std::vector<double> hh((size_t)n_fft);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n_fft; ++k) {
        long double c, sn;                         // exp(-2*pi*i*k/N)
        if (k == 0) { c = 1.0L; sn = 0.0L; }
        else if (4L * k == n_fft) { c = 0.0L; sn = -1.0L; }
        else if (2L * k == n_fft) { c = -1.0L; sn = 0.0L; }
        else if (4L * k == 3L * n_fft) { c = 0.0L; sn = 1.0L; }
        else {
            const long double a = -two_pi * (long double)k / (long double)n_fft;
            c = cosl(a);
            sn = sinl(a);
        }
        hw[k] = make_double2((double)c, (double)sn);
        hh[k] = (double)(0.5L - 0.5L * cosl(two_pi * (long double)k / (long double)n_fft));
    }
    if (!upload_table(hw, &tb.tw64) || !upload_table(hh, &tb.hann64)) {
        free_tables(tb);
        return -3;
    }
    if (is_fused_n(n_fft)) {
        // the fused f64 kernel's tables: the f32 kernel's (get_tables), evaluated in long double
        // and rounded once to double; the u8 input scale 1/128 folded into tw1 (exact)
        const int T = n_fft / 16, R3 = n_fft / 256, NP = R3 / 2;
        auto wn = [&](long num, long den, long double* c, long double* sn) {     // exp(-2 pi i num/den)
            num %= den;
            if (num == 0) { *c = 1.0L; *sn = 0.0L; }
            else if (4 * num == den) { *c = 0.0L; *sn = -1.0L; }
            else if (2 * num == den) { *c = -1.0L; *sn = 0.0L; }
            else if (4 * num == 3 * den) { *c = 0.0L; *sn = 1.0L; }
            else {
                const long double a = -two_pi * (long double)num / (long double)den;
                *c = cosl(a);
                *sn = sinl(a);
            }
        };
        std::vector<double2> h1((size_t)T * 16), h1u((size_t)T * 16), h2((size_t)NP * 16), hcs((size_t)T);
        for (int t = 0; t < T; ++t) {
            for (int s = 0; s < 16; ++s) {
                long double c, sn;
                wn((long)t * rev16h(s), n_fft, &c, &sn);
                h1[(size_t)t * 16 + s] = make_double2((double)(c * 0.0078125L), (double)(sn * 0.0078125L));
                h1u[(size_t)t * 16 + s] = make_double2((double)c, (double)sn);
            }
            const long double a = two_pi * (long double)t / (long double)n_fft;
            hcs[t] = make_double2((double)(0.5L * cosl(a)), (double)(0.5L * sinl(a)));
        }
        for (int q2 = 0; q2 < 16; ++q2) {
            int k = 0;
            for (int L = 2; L <= R3; L *= 2)
                for (int pp = 0; pp < (L >= 4 ? L / 4 : 1); ++pp) {
                    long double c, sn;       // alpha^(R3/L) * W_L^p = W_(T*L)^(q2*R3 + p*T), as (cos, sin/cos)
                    wn((long)q2 * R3 + (long)pp * T, (long)T * L, &c, &sn);
                    if (c == 0.0L) c = 1e-20L;
                    h2[(size_t)q2 * NP + k++] = make_double2((double)c, (double)(sn / c));
                }
        }
        if (!upload_table(h1, &tb.tw1_64) || !upload_table(h1u, &tb.tw1u_64) || !upload_table(h2, &tb.tw2_64) ||
            !upload_table(hcs, &tb.hann_cs64)) {
            free_tables(tb);
            return -3;
        }
        if (n_fft == 1024) {
            // spectrum_f64_1024x.hip: 1024 = 4 x 16 x 16.  Pass A (radix-16 over r on lane (p, c)) absorbs
            // the geometric part (W_64^p)^r of the twiddle W_1024^(p (c + 16 r)): the pairs of
            // alpha^(16/L) W_L^p', alpha = W_64^p, = W_(64 L)^(16 p + 64 p'), in fft_last<16>'s order.
            // The inner twiddles W_256^(c q) carry the lane constant W_1024^(p c) and the exact 1/128.
            std::vector<double2> ha((size_t)4 * 8), hb((size_t)64 * 16);
            for (int pq = 0; pq < 4; ++pq) {
                int k = 0;
                for (int L = 2; L <= 16; L *= 2)
                    for (int pp = 0; pp < (L >= 4 ? L / 4 : 1); ++pp) {
                        long double c, sn;
                        wn((long)pq * 16 + (long)pp * 64, (long)64 * L, &c, &sn);
                        if (c == 0.0L) c = 1e-20L;
                        ha[(size_t)pq * 8 + k++] = make_double2((double)c, (double)(sn / c));
                    }
            }
            for (int t = 0; t < 64; ++t) {
                const int pq = t >> 4, cc = t & 15;
                for (int s = 0; s < 16; ++s) {
                    long double c, sn;
                    wn((long)cc * (4 * rev16h(s) + pq), 1024, &c, &sn);
                    hb[(size_t)t * 16 + s] = make_double2((double)(c * 0.0078125L), (double)(sn * 0.0078125L));
                }
            }
            if (!upload_table(ha, &tb.twxa_64) || !upload_table(hb, &tb.twxb_64)) {
                free_tables(tb);
                return -3;
            }
        }
    }
    e->tables[kF64Key + n_fft] = tb;
    *out = tb;
    return 0;
}

// include/rtlws_hip.h "Streams": NULL = the engine's own non-blocking stream,
// RTLWS_STREAM_DEFAULT = HIP's legacy default stream, anything else = that stream.
hipStream_t pick_stream(rtlws_engine* e, void* stream)
{
    if (stream == RTLWS_STREAM_DEFAULT) return hipStreamLegacy;
    return stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;
}

bool desc_ok(const rtlws_spectra_desc* d)
{
    if (!d) return false;
    if (d->n_fft < 2 || d->k_avg < 1) return false;
    if (d->input < RTLWS_IN_CU8 || d->input > RTLWS_IN_RF32) return false;
    if (d->window != RTLWS_WIN_RECT && d->window != RTLWS_WIN_HANN) return false;
    if (d->output < RTLWS_OUT_POWER_SUM || d->output > RTLWS_OUT_PAYLOAD_U8) return false;
    if (d->cic_r < 0) return false;
    if (d->cic_r > 1 && d->input != RTLWS_IN_CU8) return false;
    if (!is_fused_n(d->n_fft) && d->n_fft > 8192) return false;   // direct kernel: frame in <=64 KiB LDS
    return true;
}

// spectrum_fused_v2.hip (two virtual threads per lane) for the descriptors it covers.  By
// default only K = 1 rows take it: measured +4..8 % there (rect_4096pt 0.532 -> 0.577 of the HBM
// roofline, Hann K = 1 0.471 -> 0.496) and -1..-4 % with K = 8 accumulators, where both kernels
// deliver the same points per second (DESIGN.md, configs[2]).  RTLWS_V2=0|1 forces either
// kernel for every K at engine creation (rtlws_engine_set_option(e, "v2", ...) afterwards: A/B
// runs, tests/test_v2_gpu.py).
bool use_v2(const rtlws_engine* e, int n_fft, int in_kind, int k_avg)
{
    if (!rtlws::fused_v2_kind(n_fft, in_kind)) return false;
    return e->opt.v2 >= 0 ? (e->opt.v2 != 0) : (RTLWS_V2_DEFAULT != 0 && k_avg == 1);
}

int fused_blocks(const rtlws_engine* e, int n_fft, long ngroups, int in_kind = 0, bool win = false,
                 bool kone = false, int k_avg = 0)
{
    // 4 x waves-per-SIMD wavefronts per CU, n_fft/1024 wavefronts per workgroup.
    // Persistent: each workgroup strides over the output rows.
    int per_cu = 4 * rtlws::fused_waves_per_simd(n_fft, in_kind, win, kone) / (n_fft / 1024);
    if (use_v2(e, n_fft, in_kind, k_avg)) per_cu = rtlws::v2_blocks_per_cu(n_fft);
    if (e->opt.blocks_per_cu > 0) per_cu = e->opt.blocks_per_cu;   // experiments only
    long blocks = (long)e->cu_count * per_cu;
    if (blocks > ngroups) blocks = ngroups;
    return (int)(blocks < 1 ? 1 : blocks);
}

// hipStreamWaitEvent dereferences its stream argument: the hipStreamLegacy token ((hipStream_t)1) crashes it
// (ROCm 7.2).  This library is built with the legacy default-stream semantics, where stream 0 IS that stream.
hipStream_t waitable(hipStream_t st) { return st == hipStreamLegacy ? nullptr : st; }

}  // namespace

extern "C" {

int rtlws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

int rtlws_device_pci_bus_id(int device, char* buf, int len)
{
    g_err.clear();
    if (!buf || len < 16 || device < 0 || device >= rtlws_device_count()) {
        g_err = "rtlws_device_pci_bus_id: bad device or buffer (>= 16 bytes)";
        return -1;
    }
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device), -3);
    return 0;
}

rtlws_engine* rtlws_engine_create(int device)
{
    g_err.clear();
    int n = rtlws_device_count();
    if (device < 0 || device >= n) {
        g_err = "rtlws_engine_create: no such HIP device (there is no CPU fallback)";
        return nullptr;
    }
    HIP_TRY(hipSetDevice(device), nullptr);
    rtlws_engine* e = new rtlws_engine;
    e->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        e->cu_count = prop.multiProcessorCount;
    // the only place the library reads these variables
    auto env_int = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return (v && *v) ? atoi(v) : dflt;
    };
    e->opt.v2 = env_int("RTLWS_V2", -1);
    e->opt.blocks_per_cu = env_int("RTLWS_BLOCKS_PER_CU", 0);
    e->opt.f64_fused = env_int("RTLWS_F64_FUSED", 1);
    e->opt.f64_blocks_per_cu = env_int("RTLWS_F64_BLOCKS_PER_CU", 0);
    e->opt.f64_x1024 = env_int("RTLWS_F64_X1024", 1);
    e->opt.f64_x_waves = env_int("RTLWS_F64_X_WAVES", 0);
    e->opt.cic_direct = env_int("RTLWS_CIC_DIRECT", 0) == 1;
    e->opt.cic_round = env_int("RTLWS_CIC_ROUND", 0);
    // the eight-wavefront workgroups of spectrum_f64_1024x.hip need 136 KiB of LDS: where the device cannot give
    // a workgroup that much, large batches keep the one-wavefront form (17 KiB) instead of failing
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess ||
        (size_t)lds_max < rtlws::spectra_f64_1024x_lds_bytes(8))
        e->opt.f64_x_waves8_ok = false;
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        set_err("hipStreamCreate", err);
        delete e;
        return nullptr;
    }
    return e;
}

void rtlws_engine_destroy(rtlws_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& kv : e->tables) free_tables(kv.second);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

int rtlws_engine_device(const rtlws_engine* e) { return e ? e->device : -1; }

int rtlws_engine_set_option(rtlws_engine* e, const char* name, int value)
{
    g_err.clear();
    NEED_ENGINE(e, -1);
    const std::string k = name ? name : "";
    std::lock_guard<std::mutex> lk(e->mu);
    if (k == "v2") e->opt.v2 = value < 0 ? -1 : (value != 0);
    else if (k == "blocks_per_cu") e->opt.blocks_per_cu = value > 0 ? value : 0;
    else if (k == "f64_fused") e->opt.f64_fused = value != 0;
    else if (k == "f64_blocks_per_cu") e->opt.f64_blocks_per_cu = value > 0 ? value : 0;
    else if (k == "f64_x1024") e->opt.f64_x1024 = value != 0;
    else if (k == "f64_x_waves") e->opt.f64_x_waves = (value == 1 || value == 8) ? value : 0;
    else if (k == "cic_direct") e->opt.cic_direct = value != 0;
    else if (k == "cic_round") e->opt.cic_round = (value == 1 || value == 2 || value == 4) ? value : 0;
    else {
        g_err = "rtlws_engine_set_option: unknown option '" + k + "'";
        return -1;
    }
    return 0;
}

int rtlws_engine_get_option(const rtlws_engine* e, const char* name)
{
    if (!e || !name) return -2;
    const std::string k = name;
    if (k == "v2") return e->opt.v2;
    if (k == "blocks_per_cu") return e->opt.blocks_per_cu;
    if (k == "f64_fused") return e->opt.f64_fused;
    if (k == "f64_blocks_per_cu") return e->opt.f64_blocks_per_cu;
    if (k == "f64_x1024") return e->opt.f64_x1024;
    if (k == "f64_x_waves") return e->opt.f64_x_waves;
    if (k == "cic_direct") return e->opt.cic_direct;
    if (k == "cic_round") return e->opt.cic_round;
    if (k == "cu_count") return e->cu_count;
    return -2;
}

int rtlws_engine_prepare(rtlws_engine* e, int n_fft)
{
    g_err.clear();
    rtlws_spectra_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_fft = n_fft;
    d.k_avg = 1;
    if (!e || !desc_ok(&d)) {
        g_err = "rtlws_engine_prepare: unsupported size";
        return -1;
    }
    Tables tb;
    return get_tables(e, n_fft, is_fused_n(n_fft), &tb);
}

int rtlws_engine_prepare_f64(rtlws_engine* e, int n_fft)
{
    g_err.clear();
    rtlws_spectra_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_fft = n_fft;
    d.k_avg = 1;
    if (!e || !desc_ok(&d) || n_fft > 8192) {
        g_err = "rtlws_engine_prepare_f64: unsupported size (2 <= n_fft <= 8192)";
        return -1;
    }
    Tables tb;
    if (get_tables_f64(e, n_fft, &tb) != 0) return -3;
    // Instantiations that need more than 64 KiB of LDS raise their limit with hipFuncSetAttribute, once per
    // instantiation and device: do that for every instantiation of this size NOW (launchers called with an
    // empty grid set the attribute and enqueue nothing), so that a launch -- under hipGraph capture too --
    // makes no other runtime call than the launch.
    HIP_TRY(hipSetDevice(e->device), -3);
    rtlws::SpectraParamsF64 p;
    std::memset(&p, 0, sizeof p);
    p.n_fft = n_fft;
    static const double dummy_window = 0.0;
    hipError_t err = hipSuccess;
    for (int k_avg = 1; k_avg <= 2 && err == hipSuccess; ++k_avg)
        for (int out = rtlws::OUT_SUM; out <= rtlws::OUT_PAYLOAD && err == hipSuccess; ++out)
            for (int rows_f32 = 0; rows_f32 <= 1 && err == hipSuccess; ++rows_f32) {
                p.k_avg = k_avg;
                p.out_mode = out;
                p.rows_f32 = rows_f32 && out != rtlws::OUT_PAYLOAD;
                if (n_fft == 1024 && (out == rtlws::OUT_SUM || k_avg == 1)) {
                    p.window = nullptr;
                    if (e->opt.f64_x_waves8_ok) err = rtlws::launch_spectra_f64_1024x(p, 0, 8, e->stream);
                }
                if (n_fft == 4096)
                    for (int w = 0; w <= 1 && err == hipSuccess; ++w) {
                        p.window = w ? &dummy_window : nullptr;
                        for (int in_kind : {(int)rtlws::IN_CU8, (int)rtlws::IN_CS32, (int)rtlws::IN_RF32, (int)rtlws::IN_CU8_CIC8,
                                            (int)rtlws::IN_CU8_CIC10, (int)rtlws::IN_CU8_CIC12})
                            if (err == hipSuccess) err = rtlws::launch_spectra_f64_fused_4096(p, in_kind, 0, e->stream, e->device);
                    }
            }
    if (err == hipSuccess && n_fft > 4096) {
        p.ngroups = 0;
        for (int in = RTLWS_IN_CU8; in <= RTLWS_IN_RF32 && err == hipSuccess; ++in) err = rtlws::launch_spectra_f64(p, in, e->stream, e->device);
    }
    if (err != hipSuccess) {
        set_err("rtlws_engine_prepare_f64: hipFuncSetAttribute", err);
        return -3;
    }
    return 0;
}

const char* rtlws_last_error(void) { return g_err.c_str(); }

void* rtlws_dev_alloc(rtlws_engine* e, size_t bytes)
{
    NEED_ENGINE(e, nullptr);
    void* p = nullptr;
    HIP_TRY(hipSetDevice(e->device), nullptr);
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 1), nullptr);
    return p;
}

void rtlws_dev_free(rtlws_engine* e, void* dptr)
{
    if (!dptr || !e) return;
    (void)hipSetDevice(e->device);
    (void)hipFree(dptr);
}

void* rtlws_pinned_alloc(size_t bytes)
{
    void* p = nullptr;
    // Portable | Mapped, explicitly: the zero-copy paths have kernels of ANY engine's device read
    // and write these buffers (stream rows, the drop-in staging buffers), whatever device was
    // current on the allocating thread
    HIP_TRY(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped), nullptr);
    return p;
}

void rtlws_pinned_free(void* hptr)
{
    if (hptr) (void)hipHostFree(hptr);
}

int rtlws_copy_h2d(rtlws_engine* e, void* dst, const void* src, size_t bytes, void* stream)
{
    NEED_ENGINE(e, -1);
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick_stream(e, stream)), -3);
    return 0;
}

int rtlws_copy_d2h(rtlws_engine* e, void* dst, const void* src, size_t bytes, void* stream)
{
    NEED_ENGINE(e, -1);
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick_stream(e, stream)), -3);
    return 0;
}

int rtlws_memset_dev(rtlws_engine* e, void* dst, int value, size_t bytes, void* stream)
{
    NEED_ENGINE(e, -1);
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipMemsetAsync(dst, value, bytes, pick_stream(e, stream)), -3);
    return 0;
}

int rtlws_stream_sync(rtlws_engine* e, void* stream)
{
    NEED_ENGINE(e, -1);
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipStreamSynchronize(pick_stream(e, stream)), -3);
    return 0;
}

// An event handle is {hipEvent_t, device}: the HIP event is created on the
// device of the engine it is first recorded on (a HIP event belongs to the
// device that was current when it was created, which need not be the engine's
// on a multi-GPU host), and re-created if it is later recorded on another one.
struct rtlws_event {
    hipEvent_t ev = nullptr;
    int device = -1;
    unsigned flags = hipEventDefault;
};

void* rtlws_event_create(void)
{
    return new rtlws_event;
}

void* rtlws_event_create_blocking(void)
{
    rtlws_event* x = new rtlws_event;
    x->flags = hipEventBlockingSync | hipEventDisableTiming;
    return x;
}

void* rtlws_queue_create(rtlws_engine* e)
{
    NEED_ENGINE(e, nullptr);
    HIP_TRY(hipSetDevice(e->device), nullptr);
    hipStream_t q = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking), nullptr);
    return q;
}

void rtlws_queue_destroy(rtlws_engine* e, void* queue)
{
    if (!e || !queue || queue == RTLWS_STREAM_DEFAULT) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(queue));
    (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(queue));
}

int rtlws_queue_wait_event(rtlws_engine* e, void* stream, void* ev)
{
    NEED_ENGINE(e, -1);
    rtlws_event* x = reinterpret_cast<rtlws_event*>(ev);
    if (!x || !x->ev) { g_err = "rtlws_queue_wait_event: event was never recorded"; return -1; }
    if (x->device != e->device) { g_err = "rtlws_queue_wait_event: event belongs to another device"; return -1; }
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipStreamWaitEvent(waitable(pick_stream(e, stream)), x->ev, 0), -3);
    return 0;
}

void rtlws_event_destroy(void* ev)
{
    rtlws_event* x = reinterpret_cast<rtlws_event*>(ev);
    if (!x) return;
    if (x->ev) {
        (void)hipSetDevice(x->device);
        (void)hipEventDestroy(x->ev);
    }
    delete x;
}

int rtlws_event_record(void* ev, rtlws_engine* e, void* stream)
{
    NEED_ENGINE(e, -1);
    rtlws_event* x = reinterpret_cast<rtlws_event*>(ev);
    if (!x) { g_err = "rtlws_event_record: null event"; return -1; }
    HIP_TRY(hipSetDevice(e->device), -3);
    if (x->ev && x->device != e->device) {
        (void)hipEventDestroy(x->ev);
        x->ev = nullptr;
    }
    if (!x->ev) {
        HIP_TRY(hipEventCreateWithFlags(&x->ev, x->flags), -3);
        x->device = e->device;
    }
    HIP_TRY(hipEventRecord(x->ev, pick_stream(e, stream)), -3);
    return 0;
}

int rtlws_event_sync(void* ev)
{
    rtlws_event* x = reinterpret_cast<rtlws_event*>(ev);
    if (!x || !x->ev) { g_err = "rtlws_event_sync: event was never recorded"; return -1; }
    HIP_TRY(hipSetDevice(x->device), -3);
    HIP_TRY(hipEventSynchronize(x->ev), -3);
    return 0;
}

float rtlws_event_elapsed_ms(void* start, void* stop)
{
    rtlws_event* a = reinterpret_cast<rtlws_event*>(start);
    rtlws_event* b = reinterpret_cast<rtlws_event*>(stop);
    float ms = -1.0f;
    if (!a || !b || !a->ev || !b->ev || a->device != b->device ||
        ((a->flags | b->flags) & hipEventDisableTiming)) {
        g_err = "rtlws_event_elapsed_ms: events must both be recorded, on the same device, with timing";
        return -1.0f;
    }
    HIP_TRY(hipSetDevice(a->device), -1.0f);
    HIP_TRY(hipEventSynchronize(b->ev), -1.0f);
    HIP_TRY(hipEventElapsedTime(&ms, a->ev, b->ev), -1.0f);
    return ms;
}

// Input stage of the fused kernel for a CIC factor.  For A/B experiments
// (tools/cic_fused_rates.py): option cic_direct keeps every R != 8 on the
// per-lane direct loads, cic_round = 1|2|4 forces the LDS staging depth
// where R fits it.
static int cic_in_kind(const rtlws_engine* e, int R)
{
    if (R == 8) return rtlws::IN_CU8_CIC8;
    const int force = e->opt.cic_direct ? -1 : e->opt.cic_round;
    if (force < 0) return rtlws::cicr_direct_kind(R);
    if (force == 1 || force == 2 || force == 4) {
        const int k = rtlws::cicr_lds_kind(R, force);
        if (k >= 0) return k;
    }
    return rtlws::cicr_kind(R);
}

int rtlws_spectra_kernel_kind(const rtlws_spectra_desc* d)
{
    if (!desc_ok(d)) return 0;
    return is_fused_n(d->n_fft) ? 1 : 2;
}

int rtlws_spectra_grid(rtlws_engine* e, const rtlws_spectra_desc* d, long nframes, int* blocks,
                       int* threads, int* lds_bytes)
{
    if (!e || !desc_ok(d) || nframes < 0 || nframes % d->k_avg) return -1;
    const long ngroups = nframes / d->k_avg;
    if (is_fused_n(d->n_fft)) {
        int in_kind = d->input;
        if (d->cic_r > 1) in_kind = cic_in_kind(e, d->cic_r);
        if (blocks) *blocks = fused_blocks(e, d->n_fft, ngroups, in_kind, d->window == RTLWS_WIN_HANN,
                                           d->k_avg == 1 && rtlws::fused_kone_kind(in_kind), d->k_avg);
        const bool v2 = use_v2(e, d->n_fft, in_kind, d->k_avg);
        if (threads) *threads = v2 ? d->n_fft / 32 : d->n_fft / 16;
        if (lds_bytes) *lds_bytes = v2 ? rtlws::v2_lds_bytes(d->n_fft) : rtlws::fused_lds_bytes(d->n_fft, in_kind, d->window == RTLWS_WIN_HANN);
    } else {
        if (blocks) *blocks = (int)ngroups;
        if (threads) *threads = 256;
        if (lds_bytes) *lds_bytes = (int)(sizeof(float2) * d->n_fft);
    }
    return 0;
}

int rtlws_spectra_batch(rtlws_engine* e, const rtlws_spectra_desc* d, const void* d_in,
                        long nframes, void* d_out, void* stream)
{
    g_err.clear();
    if (!e || !desc_ok(d) || !d_in || !d_out || nframes < 0 || nframes % d->k_avg) {
        g_err = "rtlws_spectra_batch: bad descriptor, pointer or frame count";
        return -1;
    }
    if (nframes == 0) return 0;
    // The kernels use 16-byte vector accesses (CIC input, N = 1024 output rows):
    // refuse pointers the hardware would fault on rather than launch.
    if ((reinterpret_cast<uintptr_t>(d_in) & 15u) || (reinterpret_cast<uintptr_t>(d_out) & 15u)) {
        g_err = "rtlws_spectra_batch: d_in and d_out must be 16-byte aligned";
        return -1;
    }
    const bool fused = is_fused_n(d->n_fft);
    Tables tb;
    if (get_tables(e, d->n_fft, fused, &tb) != 0) return -3;

    rtlws::SpectraParams p;
    std::memset(&p, 0, sizeof p);
    p.in = d_in;
    p.out = d_out;
    p.ngroups = nframes / d->k_avg;
    p.k_avg = d->k_avg;
    p.cic_r = d->cic_r > 1 ? d->cic_r : 1;
    p.n_fft = d->n_fft;
    p.out_mode = d->output;
    const bool scaled = (d->input != RTLWS_IN_RF32);
    p.tw1 = (fused && scaled) ? tb.tw1_128 : tb.tw1;
    p.tw2 = tb.tw2;
    p.in_scale = scaled ? 0.0078125f : 1.0f;
    p.window = (d->window == RTLWS_WIN_HANN) ? tb.hann : nullptr;
    p.hann_cs = tb.hann_cs;
    p.db_offset = (float)(-10.0 * std::log10((double)d->k_avg));
    // reference src/cbb_main.c:112: pow(10, gain_db/10) with C integer division
    p.lin_gain = (float)(std::pow(10.0, (double)(d->gain_db / 10)) / (double)d->k_avg);

    int in_kind = d->input;
    if (d->cic_r > 1) in_kind = cic_in_kind(e, d->cic_r);

    HIP_TRY(hipSetDevice(e->device), -3);
    hipStream_t st = pick_stream(e, stream);
    auto launch = [&](const rtlws::SpectraParams& pp, hipStream_t s) -> hipError_t {
        if (fused) {
            const int blocks = fused_blocks(e, d->n_fft, pp.ngroups, in_kind, pp.window != nullptr,
                                            d->k_avg == 1 && rtlws::fused_kone_kind(in_kind), d->k_avg);
            if (use_v2(e, d->n_fft, in_kind, d->k_avg)) return rtlws::launch_spectra_fused_v2(pp, blocks, s);
            switch (d->n_fft) {
            case 1024: return rtlws::launch_spectra_fused_1024(pp, in_kind, blocks, s);
            case 2048: return rtlws::launch_spectra_fused_2048(pp, in_kind, blocks, s);
            default: return rtlws::launch_spectra_fused_4096(pp, in_kind, blocks, s);
            }
        }
        // the direct kernel sums R bytes itself
        return rtlws::launch_spectra_direct(pp, in_kind >= rtlws::IN_CU8_CIC8 ? (int)rtlws::IN_CU8 : in_kind, s);
    };
    const hipError_t err = launch(p, st);
    if (err != hipSuccess) {
        set_err("spectra kernel launch", err);
        return -3;
    }
    return 0;
}

int rtlws_payload_from_sums(rtlws_engine* e, const float* d_sums, int n, int count, int gain_db,
                            void* d_out, void* stream)
{
    g_err.clear();
    if (!e || n < 0 || count <= 0 || (n > 0 && (!d_sums || !d_out))) {
        g_err = "rtlws_payload_from_sums: bad argument";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    const float lin = (float)(std::pow(10.0, (double)(gain_db / 10)) / (double)count);
    hipError_t err = rtlws::launch_payload(d_sums, n, lin, reinterpret_cast<uint8_t*>(d_out),
                                           pick_stream(e, stream));
    if (err != hipSuccess) {
        set_err("payload kernel launch", err);
        return -3;
    }
    return 0;
}

int rtlws_spectra_batch_f64(rtlws_engine* e, const rtlws_spectra_desc* d, const void* d_in,
                            long nframes, void* d_out, void* stream)
{
    g_err.clear();
    if (!e || !desc_ok(d) || d->n_fft > 8192 || !d_in || !d_out || nframes < 0 || nframes % d->k_avg) {
        g_err = "rtlws_spectra_batch_f64: bad descriptor (2 <= n_fft <= 8192), pointer or frame count";
        return -1;
    }
    if (nframes == 0) return 0;
    const unsigned out_align = ((d->flags & RTLWS_FLAG_ROWS_F32) || d->output == RTLWS_OUT_PAYLOAD_U8) ? 3u : 7u;
    if ((reinterpret_cast<uintptr_t>(d_in) & 7u) || (reinterpret_cast<uintptr_t>(d_out) & out_align)) {
        g_err = "rtlws_spectra_batch_f64: d_in must be 8-byte aligned, d_out 8-byte (4-byte for f32 rows and payload bytes)";
        return -1;
    }
    Tables tb;
    if (get_tables_f64(e, d->n_fft, &tb) != 0) return -3;

    rtlws::SpectraParamsF64 p;
    std::memset(&p, 0, sizeof p);
    p.in = d_in;
    p.out = d_out;
    p.ngroups = nframes / d->k_avg;
    p.k_avg = d->k_avg;
    p.cic_r = d->cic_r > 1 ? d->cic_r : 1;
    p.n_fft = d->n_fft;
    p.log2n = 0;
    if ((d->n_fft & (d->n_fft - 1)) == 0)
        for (int n = d->n_fft; n > 1; n >>= 1) ++p.log2n;
    p.out_mode = d->output;
    p.count = d->k_avg;
    p.tw = tb.tw64;
    p.window = (d->window == RTLWS_WIN_HANN) ? tb.hann64 : nullptr;
    // reference src/cbb_main.c:112: pow(10, gain_db/10) with C integer division
    p.lin_gain = std::pow(10.0, (double)(d->gain_db / 10));
    p.in_scale = (d->input != RTLWS_IN_RF32) ? 0.0078125 : 1.0;

    p.tw1f = (d->input == RTLWS_IN_RF32) ? tb.tw1u_64 : tb.tw1_64;
    p.tw2f = tb.tw2_64;
    p.hann_csf = tb.hann_cs64;
    p.rows_f32 = (d->flags & RTLWS_FLAG_ROWS_F32) && d->output != RTLWS_OUT_PAYLOAD_U8;
    p.twxa = tb.twxa_64;
    p.twxb = tb.twxb_64;

    int in_kind = d->input;
    if (d->cic_r > 1) in_kind = cic_in_kind(e, d->cic_r);
    // the fused kernel's vector accesses: 16-byte rows at N = 1024 (either row precision), 16-byte
    // input pieces on the CIC-fused kinds; anything less aligned takes the general kernel
    const bool aligned = !(reinterpret_cast<uintptr_t>(d_out) & 15u) &&
                         (in_kind < rtlws::IN_CU8_CIC8 || !(reinterpret_cast<uintptr_t>(d_in) & 15u));

    HIP_TRY(hipSetDevice(e->device), -3);
    hipStream_t st = pick_stream(e, stream);
    // the fused throughput kernel (spectrum_f64_fused.hip) where it exists; option f64_fused = 0
    // keeps everything on the row-per-workgroup kernel (A/B runs, tests)
    const bool fused = rtlws::f64_fused_kind(d->n_fft, in_kind) && e->opt.f64_fused && aligned;
    auto launch = [&](const rtlws::SpectraParamsF64& pp, hipStream_t s) -> hipError_t {
        if (!fused) return rtlws::launch_spectra_f64(pp, d->input, s, e->device);
        int per_cu = rtlws::f64_fused_blocks_per_cu(d->n_fft);
        if (e->opt.f64_blocks_per_cu > 0 && e->opt.f64_blocks_per_cu <= 2 * per_cu) per_cu = e->opt.f64_blocks_per_cu;   // experiments only
        long blocks = (long)e->cu_count * per_cu;
        if (blocks > pp.ngroups) blocks = pp.ngroups;
        // rectangular 1024-point cmplx_u8 frames: one LDS transposition instead of two
        // (its dB / payload epilogues beside K-frame accumulators would spill: those stay where they were)
        if (d->n_fft == 1024 && in_kind == rtlws::IN_CU8 && !pp.window && e->opt.f64_x1024 && pp.twxa &&
            (d->output == RTLWS_OUT_POWER_SUM || d->k_avg == 1)) {
            // batches with at least four rows per wavefront: one eight-wavefront workgroup per CU whose
            // wavefronts take the workgroup's rows one at a time (spectrum_f64_1024x.hip, WAVES)
            int waves = e->opt.f64_x_waves;
            if (waves == 0) waves = (pp.ngroups >= 32L * e->cu_count) ? 8 : 1;
            if (!e->opt.f64_x_waves8_ok) waves = 1;
            if (waves >= 8) blocks = e->cu_count;
            return rtlws::launch_spectra_f64_1024x(pp, (int)blocks, waves, s);
        }
        switch (d->n_fft) {
        case 1024: return rtlws::launch_spectra_f64_fused_1024(pp, in_kind, (int)blocks, s, e->device);
        case 2048: return rtlws::launch_spectra_f64_fused_2048(pp, in_kind, (int)blocks, s, e->device);
        default: return rtlws::launch_spectra_f64_fused_4096(pp, in_kind, (int)blocks, s, e->device);
        }
    };
    const hipError_t err = launch(p, st);
    if (err != hipSuccess) {
        set_err("f64 spectra kernel launch", err);
        return -3;
    }
    return 0;
}

int rtlws_payload_from_sums_f64(rtlws_engine* e, const double* d_sums, int n, int count, int gain_db,
                                void* d_out, void* stream)
{
    g_err.clear();
    if (!e || n < 0 || count <= 0 || (n > 0 && (!d_sums || !d_out))) {
        g_err = "rtlws_payload_from_sums_f64: bad argument";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    hipError_t err = rtlws::launch_payload_f64(d_sums, n, std::pow(10.0, (double)(gain_db / 10)), count,
                                               reinterpret_cast<uint8_t*>(d_out), pick_stream(e, stream));
    if (err != hipSuccess) {
        set_err("f64 payload kernel launch", err);
        return -3;
    }
    return 0;
}

int rtlws_welch_accumulate_f64(rtlws_engine* e, double* d_acc, const double* d_part, int n,
                               long frames_end, double* d_b, void* stream)
{
    g_err.clear();
    if (!e || n < 2 || frames_end < 0 || !d_acc || !d_part || !d_b) {
        g_err = "rtlws_welch_accumulate_f64: bad argument";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    hipError_t err = rtlws::launch_welch_accumulate(d_acc, d_part, n, frames_end, d_b, pick_stream(e, stream));
    if (err != hipSuccess) { set_err("welch accumulate kernel launch", err); return -3; }
    return 0;
}

int rtlws_welch_finish_f64(rtlws_engine* e, double* d_acc, int n, long total, double* d_b, void* stream)
{
    g_err.clear();
    if (!e || n < 2 || total < 0 || !d_acc || !d_b) {
        g_err = "rtlws_welch_finish_f64: bad argument";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    hipError_t err = rtlws::launch_welch_finish(d_acc, n, total, d_b, pick_stream(e, stream));
    if (err != hipSuccess) { set_err("welch finish kernel launch", err); return -3; }
    return 0;
}

int rtlws_cic_block_sums(rtlws_engine* e, int R, const void* d_src, long dst_len, void* d_dst,
                         void* stream)
{
    g_err.clear();
    if (!e || R < 1 || R > 128 || dst_len < 0 || (dst_len > 0 && (!d_src || !d_dst))) {
        g_err = "rtlws_cic_block_sums: bad argument (1 <= R <= 128)";
        return -1;
    }
    if ((reinterpret_cast<uintptr_t>(d_src) & 15u) || (reinterpret_cast<uintptr_t>(d_dst) & 7u)) {
        g_err = "rtlws_cic_block_sums: d_src must be 16-byte and d_dst 8-byte aligned";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    hipError_t err = rtlws::launch_cic_block_sums(R, d_src, dst_len, d_dst, pick_stream(e, stream), e->cu_count);
    if (err != hipSuccess) {
        set_err("cic kernel launch", err);
        return -3;
    }
    return 0;
}

int rtlws_fm_demod(rtlws_engine* e, const void* d_iq, long len, const float* d_prev_in,
                   float* d_prev_out, float* d_out, void* stream)
{
    g_err.clear();
    if (!e || len < 0 || !d_prev_in || !d_prev_out || d_prev_in == d_prev_out ||
        (len > 0 && (!d_iq || !d_out))) {
        g_err = "rtlws_fm_demod: bad argument";
        return -1;
    }
    if ((reinterpret_cast<uintptr_t>(d_iq) & 7u) || (reinterpret_cast<uintptr_t>(d_out) & 3u)) {
        g_err = "rtlws_fm_demod: d_iq must be 8-byte and d_out 4-byte aligned";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    if (len == 0) {   // nothing to demodulate: the carried phase passes through
        HIP_TRY(hipMemcpyAsync(d_prev_out, d_prev_in, sizeof(float), hipMemcpyDeviceToDevice,
                               pick_stream(e, stream)), -3);
        return 0;
    }
    hipError_t err = rtlws::launch_fm_demod(d_iq, len, d_prev_in, d_prev_out, d_out, pick_stream(e, stream), e->cu_count);
    if (err != hipSuccess) {
        set_err("fm_demod kernel launch", err);
        return -3;
    }
    return 0;
}

// ---- shader-clock probe -------------------------------------------------------------------
struct rtlws_clock_probe {
    rtlws_engine* e = nullptr;
    hipStream_t q = nullptr;            // its own queue: runs beside whatever the caller times
    int* stop = nullptr;                // pinned, device-visible
    unsigned long long* out = nullptr;  // pinned: {shader clocks, 100 MHz ticks, polls, resident flag}
};

int rtlws_clock_stamp(rtlws_engine* e, unsigned long long* d_out, int slots, void* stream)
{
    g_err.clear();
    NEED_ENGINE(e, -1);
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 7u) || slots < 1 || slots > 65536) {
        g_err = "rtlws_clock_stamp: d_out must be an 8-byte aligned device pointer to slots x 4 64-bit words, 1 <= slots <= 65536";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    const hipError_t err = rtlws::launch_clock_stamp(d_out, slots, pick_stream(e, stream));
    if (err != hipSuccess) { set_err("clock stamp kernel launch", err); return -3; }
    return 0;
}

void* rtlws_clock_probe_start(rtlws_engine* e)
{
    g_err.clear();
    NEED_ENGINE(e, nullptr);
    HIP_TRY(hipSetDevice(e->device), nullptr);
    rtlws_clock_probe* p = new rtlws_clock_probe;
    p->e = e;
    hipError_t err = hipStreamCreateWithFlags(&p->q, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void**>(&p->stop), sizeof(int), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void**>(&p->out), 4 * sizeof(unsigned long long), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (err == hipSuccess) {
        *p->stop = 0;
        p->out[0] = p->out[1] = p->out[2] = p->out[3] = 0;
        // at most 2^24 polls of ~0.5 us: the wavefront leaves after ~10 s whatever the host does
        err = rtlws::launch_clock_probe(p->stop, p->out, 1 << 24, p->q);
        // The first launch on a new queue sets the queue up (about a millisecond, during which launches
        // on other queues wait): return only once the wavefront is resident, so none of that lands in
        // the interval the caller is about to time.  Bounded: ~2 s, then the probe is used as it is.
        for (long spin = 0; err == hipSuccess && spin < 2000000L; ++spin) {
            if (__atomic_load_n(&p->out[3], __ATOMIC_ACQUIRE)) break;
            struct timespec ts = {0, 1000};
            nanosleep(&ts, nullptr);
        }
    }
    if (err != hipSuccess) {
        set_err("rtlws_clock_probe_start", err);
        if (p->q) (void)hipStreamDestroy(p->q);
        if (p->stop) (void)hipHostFree(p->stop);
        if (p->out) (void)hipHostFree(p->out);
        delete p;
        return nullptr;
    }
    return p;
}

void rtlws_clock_probe_signal(void* probe)
{
    rtlws_clock_probe* p = reinterpret_cast<rtlws_clock_probe*>(probe);
    if (p) __atomic_store_n(p->stop, 1, __ATOMIC_RELEASE);
}

int rtlws_clock_probe_signal_on_stream(void* probe, void* stream)
{
    g_err.clear();
    rtlws_clock_probe* p = reinterpret_cast<rtlws_clock_probe*>(probe);
    if (!p) { g_err = "rtlws_clock_probe_signal_on_stream: null probe"; return -1; }
    HIP_TRY(hipSetDevice(p->e->device), -3);
    // the command processor writes the flag when it reaches this packet, i.e. once everything enqueued on
    // `stream` before it has completed: no host round trip between the last launch and the probe's exit
    HIP_TRY(hipStreamWriteValue32(pick_stream(p->e, stream), p->stop, 1, 0), -3);
    return 0;
}

int rtlws_clock_probe_stop(void* probe, double* sclk_ghz, double* seconds)
{
    g_err.clear();
    rtlws_clock_probe* p = reinterpret_cast<rtlws_clock_probe*>(probe);
    if (!p) { g_err = "rtlws_clock_probe_stop: null probe"; return -1; }
    int rc = 0;
    (void)hipSetDevice(p->e->device);
    __atomic_store_n(p->stop, 1, __ATOMIC_RELEASE);
    hipError_t err = hipStreamSynchronize(p->q);
    if (err != hipSuccess) { set_err("rtlws_clock_probe_stop", err); rc = -3; }
    const double clocks = (double)p->out[0], ticks = (double)p->out[1];
    if (rc == 0 && ticks <= 0.0) { g_err = "rtlws_clock_probe_stop: the probe recorded no interval"; rc = -3; }
    if (rc == 0) {
        if (sclk_ghz) *sclk_ghz = clocks / ticks * 0.1;       // ticks are 10 ns
        if (seconds) *seconds = ticks * 1e-8;
    }
    (void)hipStreamDestroy(p->q);
    (void)hipHostFree(p->stop);
    (void)hipHostFree(p->out);
    delete p;
    return rc;
}

int rtlws_copy_d2d(rtlws_engine* e, void* dst, const void* src, size_t bytes, void* stream)
{
    NEED_ENGINE(e, -1);
    HIP_TRY(hipSetDevice(e->device), -3);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick_stream(e, stream)), -3);
    return 0;
}

int rtlws_halfband(rtlws_engine* e, const float* d_x, float* d_y, long out_len, void* stream)
{
    g_err.clear();
    if (!e || out_len < 0 || (out_len > 0 && (!d_x || !d_y))) {
        g_err = "rtlws_halfband: bad argument";
        return -1;
    }
    HIP_TRY(hipSetDevice(e->device), -3);
    hipError_t err = rtlws::launch_halfband(d_x, d_y, out_len, pick_stream(e, stream), e->cu_count);
    if (err != hipSuccess) {
        set_err("halfband kernel launch", err);
        return -3;
    }
    return 0;
}

}  // extern "C"
