// fake_rtlws_hip.cpp -- TEST-ONLY stand-in for librtlws_hip.so on the CPU.  NEVER part of the product: it is
// built by tests/test_host_sanitizers_cpu.py into a scratch directory, never into rtl-ws_amd/lib.
//
// Why it exists: the threaded host layer -- the ring walk and condition variables of
// rtl-ws_amd/host/stream_gpu.c, the two-slot hand-off of cbb_gpu.c, the shard threads and command mailbox of
// multi_batch.c, rf_decimator_set_parameters from a second thread (reference src/main.c:154) -- only executes
// its real logic when copies, launches and events SUCCEED, i.e. on a GPU box, where no sanitizer runs (GPU ASan
// is not available on the pool).  This file implements include/rtlws_hip.h with host memory and threads so that
// the same host sources run that logic here under -fsanitize=thread and -fsanitize=address,undefined:
//   * a stream / queue is a worker thread with a FIFO of closures; every closure starts after a random delay, so
//     completion order across queues, and the time between enqueue and completion, vary from run to run;
//   * "device" and "pinned" memory are malloc'd; copies are memcpy on the queue's thread;
//   * an event completes when its queue reaches it; rtlws_event_sync sleeps on a condition variable;
//   * the transforms are the f64 oracle's (oracle/rtlws_oracle.c is test infrastructure, as this file is), so the
//     rows a test receives can be checked against a direct oracle call.
// Semantics kept from the real shim: NULL stream = the engine's own queue; work on one queue runs in order;
// nothing is ordered across queues except through events; every entry point tolerates NULL handles on free.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "rtlws_hip.h"

extern "C" {
#include "rtlws_oracle.h"
}

namespace {

thread_local std::string g_err;
std::atomic<long> g_live_allocs{0}, g_live_events{0}, g_live_engines{0}, g_live_queues{0};
std::atomic<int> g_fail_after{-1};        // FAKE_HIP_FAIL_AFTER=n: the n-th spectra launch from now fails with -3

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

struct Queue {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv, cv_idle;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::mt19937 rng;
    int max_delay_us;

    explicit Queue(unsigned seed) : rng(seed), max_delay_us(env_int("FAKE_HIP_MAX_DELAY_US", 300))
    {
        th = std::thread([this] { run(); });
    }
    ~Queue()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            int delay;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;           // stop and drained
                f = std::move(q.front());
                q.pop_front();
                busy = true;
                delay = max_delay_us > 0 ? (int)(rng() % (unsigned)max_delay_us) : 0;
            }
            if (delay) std::this_thread::sleep_for(std::chrono::microseconds(delay));
            f();
            {
                std::lock_guard<std::mutex> lk(mu);
                busy = false;
            }
            cv_idle.notify_all();
        }
    }
    void push(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
        }
        cv.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [this] { return q.empty() && !busy; });
    }
};

struct Event {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long recorded = 0, completed = 0;
    std::chrono::steady_clock::time_point when;
};

std::atomic<unsigned> g_seed{12345};

}  // namespace

struct rtlws_engine {
    int device;
    Queue own, legacy;
    std::mutex mu;
    std::map<std::string, int> opt;
    explicit rtlws_engine(int d) : device(d), own(g_seed.fetch_add(7)), legacy(g_seed.fetch_add(7)) {}
};

namespace {

Queue* pick(rtlws_engine* e, void* stream)
{
    if (stream == RTLWS_STREAM_DEFAULT) return &e->legacy;
    return stream ? static_cast<Queue*>(stream) : &e->own;
}

bool desc_ok(const rtlws_spectra_desc* d)
{
    if (!d || d->n_fft < 2 || d->k_avg < 1) return false;
    if (d->input < RTLWS_IN_CU8 || d->input > RTLWS_IN_RF32) return false;
    if (d->window != RTLWS_WIN_RECT && d->window != RTLWS_WIN_HANN) return false;
    if (d->output < RTLWS_OUT_POWER_SUM || d->output > RTLWS_OUT_PAYLOAD_U8) return false;
    if (d->cic_r < 0 || (d->cic_r > 1 && d->input != RTLWS_IN_CU8)) return false;
    return d->n_fft <= 8192;
}

// one batch through the oracle: f64 sums, then the epilogue the descriptor asks for, stored as `elem` bytes each
void transform(const rtlws_spectra_desc d, const void* in, long nframes, void* out, int elem)
{
    const int N = d.n_fft, K = d.k_avg;
    const long rows = nframes / K;
    std::vector<double> win;
    if (d.window == RTLWS_WIN_HANN) {
        win.resize(N);
        for (int i = 0; i < N; i++) win[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / N);
    }
    const double* w = win.empty() ? nullptr : win.data();
    std::vector<double> sums((size_t)rows * N, 0.0);
    if (d.input == RTLWS_IN_CU8) {
        if (d.cic_r > 1) orc_batch_spectra_cic_u8(N, K, d.cic_r, nframes, static_cast<const uint8_t*>(in), w, sums.data(), 1);
        else orc_batch_spectra_u8(N, K, nframes, static_cast<const uint8_t*>(in), w, sums.data(), 1);
    } else {
        for (long r = 0; r < rows; r++)
            for (int k = 0; k < K; k++) {
                const size_t f = (size_t)(r * K + k);
                if (d.input == RTLWS_IN_CS32) orc_spectrum_add_cmplx_s32(N, static_cast<const int32_t*>(in) + f * 2 * N, w, &sums[(size_t)r * N], N);
                else orc_spectrum_add_real_f32(N, static_cast<const float*>(in) + f * N, w, &sums[(size_t)r * N], N);
            }
    }
    for (long r = 0; r < rows; r++) {
        const double* ps = &sums[(size_t)r * N];
        if (d.output == RTLWS_OUT_PAYLOAD_U8) {
            orc_spectrum_payload(N, ps, K, d.gain_db, static_cast<uint8_t*>(out) + (size_t)r * N);
            continue;
        }
        for (int i = 0; i < N; i++) {
            const double v = d.output == RTLWS_OUT_MEAN_DB ? 10.0 * std::log10(ps[i] / K) : ps[i];
            if (elem == 8) static_cast<double*>(out)[(size_t)r * N + i] = v;
            else static_cast<float*>(out)[(size_t)r * N + i] = (float)v;
        }
    }
}

int launch(rtlws_engine* e, const rtlws_spectra_desc* d, const void* in, long nframes, void* out, void* stream, int elem)
{
    g_err.clear();
    if (!e || !desc_ok(d) || !in || !out || nframes < 0 || nframes % d->k_avg) {
        g_err = "fake shim: bad descriptor, pointer or frame count";
        return -1;
    }
    if (nframes == 0) return 0;
    int fa = g_fail_after.load();
    while (fa >= 0 && !g_fail_after.compare_exchange_weak(fa, fa - 1)) {}
    if (fa == 0) {
        g_err = "fake shim: injected launch failure";
        return -3;
    }
    const rtlws_spectra_desc dd = *d;
    pick(e, stream)->push([=] { transform(dd, in, nframes, out, elem); });
    return 0;
}

}  // namespace

extern "C" {

// test hooks (not in rtlws_hip.h)
long fake_hip_live_objects(void) { return g_live_allocs + g_live_events + g_live_engines + g_live_queues; }
void fake_hip_fail_after(int n) { g_fail_after = n; }

int rtlws_device_count(void) { return env_int("FAKE_HIP_DEVICES", 2); }

int rtlws_device_pci_bus_id(int device, char* buf, int len)
{
    if (!buf || len < 16 || device < 0 || device >= rtlws_device_count()) return -1;
    snprintf(buf, (size_t)len, "0000:%02x:00.0", 0x05 + 0x10 * device);
    return 0;
}

rtlws_engine* rtlws_engine_create(int device)
{
    g_err.clear();
    if (device < 0 || device >= rtlws_device_count()) {
        g_err = "fake shim: no such device";
        return nullptr;
    }
    ++g_live_engines;
    return new rtlws_engine(device);
}

void rtlws_engine_destroy(rtlws_engine* e)
{
    if (!e) return;
    e->own.drain();
    e->legacy.drain();
    delete e;
    --g_live_engines;
}

int rtlws_engine_device(const rtlws_engine* e) { return e ? e->device : -1; }

int rtlws_engine_set_option(rtlws_engine* e, const char* name, int value)
{
    if (!e || !name) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    e->opt[name] = value;
    return 0;
}

int rtlws_engine_get_option(const rtlws_engine* e, const char* name)
{
    if (!e || !name) return -2;
    rtlws_engine* m = const_cast<rtlws_engine*>(e);
    std::lock_guard<std::mutex> lk(m->mu);
    if (std::string(name) == "cu_count") return 256;
    auto it = m->opt.find(name);
    return it == m->opt.end() ? -2 : it->second;
}

int rtlws_engine_prepare(rtlws_engine* e, int n_fft) { return (e && n_fft >= 2 && n_fft <= 8192) ? 0 : -1; }
int rtlws_engine_prepare_f64(rtlws_engine* e, int n_fft) { return rtlws_engine_prepare(e, n_fft); }
const char* rtlws_last_error(void) { return g_err.c_str(); }

void* rtlws_dev_alloc(rtlws_engine* e, size_t bytes)
{
    if (!e) return nullptr;
    ++g_live_allocs;
    return malloc(bytes ? bytes : 1);
}
void rtlws_dev_free(rtlws_engine* e, void* p)
{
    if (!p || !e) return;
    free(p);
    --g_live_allocs;
}
void* rtlws_pinned_alloc(size_t bytes)
{
    ++g_live_allocs;
    return malloc(bytes ? bytes : 1);
}
void rtlws_pinned_free(void* p)
{
    if (!p) return;
    free(p);
    --g_live_allocs;
}

static int copy(rtlws_engine* e, void* dst, const void* src, size_t bytes, void* stream)
{
    if (!e || (bytes && (!dst || !src))) return -1;
    pick(e, stream)->push([=] { memcpy(dst, src, bytes); });
    return 0;
}
int rtlws_copy_h2d(rtlws_engine* e, void* d, const void* s, size_t n, void* st) { return copy(e, d, s, n, st); }
int rtlws_copy_d2h(rtlws_engine* e, void* d, const void* s, size_t n, void* st) { return copy(e, d, s, n, st); }
int rtlws_copy_d2d(rtlws_engine* e, void* d, const void* s, size_t n, void* st) { return copy(e, d, s, n, st); }
int rtlws_memset_dev(rtlws_engine* e, void* dst, int value, size_t bytes, void* stream)
{
    if (!e || (bytes && !dst)) return -1;
    pick(e, stream)->push([=] { memset(dst, value, bytes); });
    return 0;
}
int rtlws_stream_sync(rtlws_engine* e, void* stream)
{
    if (!e) return -1;
    pick(e, stream)->drain();
    return 0;
}

void* rtlws_queue_create(rtlws_engine* e)
{
    if (!e) return nullptr;
    ++g_live_queues;
    return new Queue(g_seed.fetch_add(7));
}
void rtlws_queue_destroy(rtlws_engine* e, void* q)
{
    if (!e || !q) return;
    static_cast<Queue*>(q)->drain();
    delete static_cast<Queue*>(q);
    --g_live_queues;
}

void* rtlws_event_create(void)
{
    ++g_live_events;
    return new Event;
}
void* rtlws_event_create_blocking(void) { return rtlws_event_create(); }
void rtlws_event_destroy(void* ev)
{
    if (!ev) return;
    delete static_cast<Event*>(ev);
    --g_live_events;
}
int rtlws_event_record(void* ev, rtlws_engine* e, void* stream)
{
    if (!ev || !e) return -1;
    Event* x = static_cast<Event*>(ev);
    unsigned long gen;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        gen = ++x->recorded;
    }
    pick(e, stream)->push([x, gen] {
        // notify under the lock: a waiter may destroy the event as soon as it has seen `completed`
        std::lock_guard<std::mutex> lk(x->mu);
        if (gen > x->completed) x->completed = gen;
        x->when = std::chrono::steady_clock::now();
        x->cv.notify_all();
    });
    return 0;
}
int rtlws_event_sync(void* ev)
{
    if (!ev) return -3;
    Event* x = static_cast<Event*>(ev);
    std::unique_lock<std::mutex> lk(x->mu);
    const unsigned long want = x->recorded;
    x->cv.wait(lk, [&] { return x->completed >= want; });
    return 0;
}
float rtlws_event_elapsed_ms(void* a, void* b)
{
    if (!a || !b) return -1.0f;
    rtlws_event_sync(b);
    Event *x = static_cast<Event*>(a), *y = static_cast<Event*>(b);
    std::chrono::steady_clock::time_point ta, tb;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        ta = x->when;
    }
    {
        std::lock_guard<std::mutex> lk(y->mu);
        tb = y->when;
    }
    return std::chrono::duration<float, std::milli>(tb - ta).count();
}
int rtlws_queue_wait_event(rtlws_engine* e, void* stream, void* ev)
{
    if (!e || !ev) return -1;
    Event* x = static_cast<Event*>(ev);
    unsigned long want;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        want = x->recorded;
    }
    pick(e, stream)->push([x, want] {
        std::unique_lock<std::mutex> lk(x->mu);
        x->cv.wait(lk, [&] { return x->completed >= want; });
    });
    return 0;
}

int rtlws_spectra_kernel_kind(const rtlws_spectra_desc* d)
{
    if (!desc_ok(d)) return 0;
    return (d->n_fft == 1024 || d->n_fft == 2048 || d->n_fft == 4096) ? 1 : 2;
}
int rtlws_spectra_grid(rtlws_engine* e, const rtlws_spectra_desc* d, long nframes, int* blocks, int* threads, int* lds)
{
    if (!e || !desc_ok(d) || nframes < 0) return -1;
    if (blocks) *blocks = 1;
    if (threads) *threads = 64;
    if (lds) *lds = 0;
    return 0;
}

int rtlws_spectra_batch(rtlws_engine* e, const rtlws_spectra_desc* d, const void* in, long nframes, void* out, void* stream)
{
    return launch(e, d, in, nframes, out, stream, 4);
}
int rtlws_spectra_batch_f64(rtlws_engine* e, const rtlws_spectra_desc* d, const void* in, long nframes, void* out, void* stream)
{
    return launch(e, d, in, nframes, out, stream, (d && (d->flags & RTLWS_FLAG_ROWS_F32)) ? 4 : 8);
}

int rtlws_payload_from_sums(rtlws_engine* e, const float* sums, int n, int count, int gain_db, void* out, void* stream)
{
    if (!e || n < 0 || count <= 0 || (n > 0 && (!sums || !out))) return -1;
    pick(e, stream)->push([=] {
        std::vector<double> ps(sums, sums + n);
        orc_spectrum_payload(n, ps.data(), count, gain_db, static_cast<uint8_t*>(out));
    });
    return 0;
}
int rtlws_payload_from_sums_f64(rtlws_engine* e, const double* sums, int n, int count, int gain_db, void* out, void* stream)
{
    if (!e || n < 0 || count <= 0 || (n > 0 && (!sums || !out))) return -1;
    pick(e, stream)->push([=] { orc_spectrum_payload(n, sums, count, gain_db, static_cast<uint8_t*>(out)); });
    return 0;
}
int rtlws_welch_accumulate_f64(rtlws_engine* e, double* acc, const double* part, int n, long frames_end, double* b, void* stream)
{
    if (!e || n < 2 || frames_end < 0 || !acc || !part || !b) return -1;
    pick(e, stream)->push([=] {
        for (int i = 0; i < n; i++) acc[i] += part[i];
        *b += part[n / 2 - 1] * (double)frames_end;
    });
    return 0;
}
int rtlws_welch_finish_f64(rtlws_engine* e, double* acc, int n, long total, double* b, void* stream)
{
    if (!e || n < 2 || total < 0 || !acc || !b) return -1;
    pick(e, stream)->push([=] {
        acc[n / 2] += (double)total * acc[n / 2 - 1] - *b;
        *b = 0.0;
    });
    return 0;
}

int rtlws_cic_block_sums(rtlws_engine* e, int R, const void* src, long dst_len, void* dst, void* stream)
{
    if (!e || R < 1 || dst_len < 0 || (dst_len && (!src || !dst))) return -1;
    pick(e, stream)->push([=] {
        const uint8_t* s = static_cast<const uint8_t*>(src);
        int32_t* d = static_cast<int32_t*>(dst);
        for (long m = 0; m < dst_len; m++) {
            int32_t re = 0, im = 0;
            for (int k = 0; k < R; k++) {
                re += (int32_t)s[2 * (m * R + k)] - 128;
                im += (int32_t)s[2 * (m * R + k) + 1] - 128;
            }
            d[2 * m] = re;
            d[2 * m + 1] = im;
        }
    });
    return 0;
}
int rtlws_halfband(rtlws_engine* e, const float* x, float* y, long out_len, void* stream)
{
    if (!e || out_len < 0 || (out_len && (!x || !y))) return -1;
    pick(e, stream)->push([=] {
        // d_x holds the 10 history samples in front: the oracle wants them as its delay line
        float delay[10];
        memcpy(delay, x, sizeof delay);
        orc_halfband_decimate(x + 10, y, (int)out_len, delay);
    });
    return 0;
}
int rtlws_fm_demod(rtlws_engine* e, const void* iq, long len, const float* prev_in, float* prev_out, float* out, void* stream)
{
    if (!e || len < 0 || !prev_in || !prev_out || prev_in == prev_out || (len && (!iq || !out))) return -1;
    pick(e, stream)->push([=] {
        float prev = *prev_in;
        orc_fm_demod(static_cast<const int32_t*>(iq), (int)len, &prev, out);
        *prev_out = prev;
    });
    return 0;
}

struct Probe { std::chrono::steady_clock::time_point t0; };
void* rtlws_clock_probe_start(rtlws_engine* e) { return e ? new Probe{std::chrono::steady_clock::now()} : nullptr; }
void rtlws_clock_probe_signal(void*) {}
int rtlws_clock_probe_signal_on_stream(void* p, void*) { return p ? 0 : -1; }
int rtlws_clock_probe_stop(void* p, double* ghz, double* seconds)
{
    if (!p) return -1;
    Probe* x = static_cast<Probe*>(p);
    if (ghz) *ghz = 2.0;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - x->t0).count();
    delete x;
    return 0;
}

}  // extern "C"
