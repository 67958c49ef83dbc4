// C ABI of libgpx.so (see include/gpx.h). Owns the device state of one exact GP
// and sequences the HIP kernels of kmat.hip / chol.hip / vec.hip / gemm_f64.hip
// on the handle's stream. One host synchronisation per call, at the end, when
// the handful of result doubles is copied back.

#include "gpx_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

// ---- error string ------------------------------------------------------------
static thread_local char g_err[1024] = "";

void gpx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- handle ------------------------------------------------------------------
enum { T_BUILD = 0, T_POTRF, T_TRSV, T_TRTRI, T_TRMV, T_LAUUM, T_TRACE, T_SCALARS,
       T_POST_BUILD, T_POST_SOLVE };
static const char *kTimerNames[GPX_NTIMERS] = {
    "kernel_build", "potrf", "trsv", "trtri", "trmv", "lauum", "trace_grad",
    "scalars", "posterior_build", "posterior_solve"};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need)
    {
        if (need <= bytes) return 0;
        if (p) GPX_HIP(hipFree(p));
        p = nullptr;
        bytes = 0;
        GPX_HIP(hipMalloc(&p, need));
        bytes = need;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct gpx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // resident data (GP.add_data)
    int n = 0, d = 0, np = 0, ld = 0;
    int cap = 0;                   // rows the matrices are allocated for (>= np): room
                                   // for gpx_exact_append to open new 128-blocks
    long data_version = 0;         // bumped by gpx_set_data (twin refresh)
    DevBuf X, y, Xf32;
    // factorisation state
    DevBuf A, W, Kinv, r, a, alpha, scalars, acc, partial, info, gv_part, pctl;
    KParams kp;
    double log_sn = 0, mean = 0;
    bool have_factor = false, have_inverse = false;
    bool w_complete = false;   // W holds the whole R^-1 (not just the diagonal blocks)
    bool kinv_ready = false;   // Kinv = (R^T R)^-1 came out of the factorisation
    bool kinv_pending = false; // ... and its last update has not been joined yet (enqueue_grad does)
    int gate_total[2] = {0, 0};    // moves of the panel gates enqueued so far (chol.hip)
    bool lz_enqueued = false;      // the scalar terms of this evaluation are already queued
    bool batch_la = false;         // this batch runs its members with look-ahead (large N)
    bool no_panel = false;         // safe mode (gpx_set_safe_mode): diagonal blocks by recursion
                                   // down to the 128-tiles, no task-queue launches
    bool in_batch = false;         // this context runs members of a batch that keeps several
                                   // contexts in flight (set by the batch entry points)
    int posterior_calls = 0;       // since the last factorisation (posterior_impl)
    double lZ = 0;
    // posterior / api scratch
    DevBuf Ks, KsT, Xs, mu, s2, post_part, t0, t1, t2, split, gpart;
    int64_t bench_n = 0;
    // results of the evaluation in flight land in pinned host memory
    double *hres = nullptr;        // [0..2] scalars, [4..] trace accumulators
    int *hinfo = nullptr;
    bool pending_grad = false;
    // second context (own stream + workspace) used by gpx_loglik_batch to keep
    // two independent evaluations in flight on this GPU
    gpx_ctx *twin = nullptr;
    // member-batched evaluation of batches (group.hip), created on first use
    GpxGroups *groups = nullptr;
    // look-ahead of the factorisation (chol.hip): diagonal blocks on a high-priority
    // stream, the left half of the inverse tree on a low-priority one
    hipStream_t crit = nullptr, crit_only = nullptr, aux = nullptr, bulk = nullptr;
    int bulk_slots = 0;
    bool stream_borrowed = false;  // batch context: `stream` belongs to the device's pool
    bool bulk_borrowed = false;    // ... and `bulk` is that stream
    std::vector<hipStream_t> chain_streams;   // ... and the streams of the contexts before it
    int twin_index = 0;            // position in the chain of batch contexts (0: a handle)
    hipEvent_t la_events[GPX_LA_EVENTS] = {};
    // timing
    bool timing = false;
    hipEvent_t ev[GPX_NTIMERS + 1] = {};
    bool ev_used[GPX_NTIMERS + 1] = {};
    double ms[GPX_NTIMERS] = {};

    DenseWs ws() const
    {
        DenseWs w;
        w.A = A.as<double>();
        w.W = W.as<double>();
        w.Kinv = Kinv.as<double>();
        w.np = np;
        w.ld = ld;
        w.info = info.as<int>();
        w.pctl = no_panel ? nullptr : pctl.as<int>();
        w.gate_total = const_cast<int *>(gate_total);
        // one evaluation at a time: hide the diagonal-block chain under its own
        // trailing updates. Several evaluations in flight (batch entry points) hide
        // it under each other and keep to one stream each.
        // it under each other and keep to one stream each -- except large ones (batch_la,
        // set by gpx_loglik_batch), which do both.
        // (a property of THIS context, constant over an evaluation: until the end of round 4
        // the device-wide count of running batches was read here, at every call -- a batch
        // started by another host thread between an update that had deferred its last K^-1
        // product to the third stream and the gradient stage that joins it made the join
        // find no streams and the trace terms read an unfinished K^-1: wrong gradients,
        // found by tools/soak_threads.py big)
        if (crit && (!in_batch || batch_la)) {
            w.crit = crit;
            w.crit_only = crit_only;
            w.aux = aux;
            w.bulk = bulk;
            w.bulk_slots = bulk_slots;
            w.events = const_cast<hipEvent_t *>(la_events);
        }
        return w;
    }
};

static inline int round_up(int64_t x, int m) { return (int)((x + m - 1) / m * m); }

// Row stride of the np x np device matrices. A power-of-two stride puts the
// same column of every row on one HBM channel; GPX_LDPAD doubles (default 32 =
// 256 B) rotate consecutive rows over the channels.
static int ld_for(int np)
{
    static const int pad = [] {
        const char *e = getenv("GPX_LDPAD");
        const int v = e ? atoi(e) : 32;
        return v < 0 || v % 2 ? 32 : v;
    }();
    return np + pad;
}

#define CHECK_H(h)                                                             \
    do {                                                                       \
        if (!(h)) {                                                            \
            gpx_set_error("null handle");                                      \
            return -1;                                                         \
        }                                                                      \
        GPX_HIP(hipSetDevice((h)->device));                                    \
    } while (0)

// stage timer: measures from the previous tick to this one
struct StageClock {
    gpx_ctx *h;
    hipEvent_t prev;
    explicit StageClock(gpx_ctx *h_) : h(h_), prev(nullptr)
    {
        if (h->timing) {
            // stages that this call does not tick read 0, not what an earlier call left
            // (with the K^-1 update deferred, trsv and trmv are part of the potrf stage)
            for (int i = 0; i < GPX_NTIMERS; ++i) {
                h->ev_used[i] = false;
                h->ms[i] = 0.0;
            }
            (void)hipEventRecord(h->ev[GPX_NTIMERS], h->stream);
        }
    }
    void tick(int stage)
    {
        if (!h->timing) return;
        (void)hipEventRecord(h->ev[stage], h->stream);
        h->ev_used[stage] = true;
        if (count < (int)(sizeof order / sizeof *order)) order[count++] = stage;
    }
    void collect()
    {
        if (!h->timing) return;
        hipEvent_t last = h->ev[GPX_NTIMERS];
        for (int i = 0; i < count; ++i) {
            float t = 0;
            (void)hipEventSynchronize(h->ev[order[i]]);
            (void)hipEventElapsedTime(&t, last, h->ev[order[i]]);
            h->ms[order[i]] = t;
            last = h->ev[order[i]];
        }
    }
    int order[GPX_NTIMERS + 2];
    int count = 0;
};

extern "C" {

int gpx_version(void) { return GPX_VERSION; }

const char *gpx_last_error(void) { return g_err; }

int gpx_device_count(int *count)
{
    if (!count) {
        gpx_set_error("gpx_device_count: null");
        return -1;
    }
    *count = 0;
    GPX_HIP(hipGetDeviceCount(count));
    return 0;
}

// set while ensure_twin creates a batch context: 1 = plain, 2 = small-problem variant
static thread_local int g_creating_twin = 0;
static thread_local int g_twin_index = 0;      // position of the context in its chain (1, 2, ...)

// spins for `ticks` of the 100-MHz wall clock (the queue probe below; GPX_TEST_HOLD_BUILD_US)
__global__ void hold_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ---- streams of batch contexts ------------------------------------------------------
// Batch members run one per context, each on its context's stream, and they only run side
// by side if those streams sit on different hardware queues: the runtime hands a stream
// the least-used of its few queues when the stream is first needed, kernels of two
// streams that share a queue run one after the other, and which streams share depends on
// everything the process created and used before. Round 2 met this as "194 or 240
// evals/s for 64 thetas at N = 8192, depending on whether an unused CU-masked queue had
// been created in front of a twin's stream"; round 3 first fixed the creation order of a
// per-device pool and still found 188 instead of 250 evals/s whenever single evaluations
// (which use the look-ahead streams) had run on the handle before its first batch
// (tools/r03_exp38.sh): the second member then shared the first member's queue.
// Now the layout is MEASURED instead of arranged: the twin streams of a device still come
// from one pool, created once, and a context takes the first pool stream that does not
// share a queue with any stream of the contexts before it in its chain. The probe: a
// kernel that spins for 0.3 ms on one stream, an empty kernel on the other -- the empty
// one ends first unless it had to queue behind the spinner. A few probes of 0.3 ms, once
// per batch context. (Two handles batching on one device from two threads at the same
// time may pick the same pool streams: correct, their members then queue behind each
// other.) A second probe (stream_pair_cost below) finds the pairs of queues that are not
// shared but run badly side by side (GPX_TWIN_LOG=1 prints the costs seen), and the pool's
// streams have hardware queues of their own (twin_pool_make), so that a good one exists
// whatever the process did before. Measured and dropped: making every queue up front, in one order, at handle
// creation (this stream, the pool, each touched once): uniform -- and uniformly worse, the
// streams made later then share queues with the touched ones (64 thetas 202-207 evals/s
// in every order, one evaluation at N = 4096 2.42 -> 2.82 ms, the metric batch 14.4 ->
// 11.8 evals/s).
// ---- CU-masked streams are never destroyed ------------------------------------------
// hipStreamDestroy of a stream made with hipExtStreamCreateWithCUMask did not return in 3 of
// about 90 teardowns at the end of round 4 (always in the teardown of a handle, never in a
// kernel). Every masked stream the library makes -- the opt-in CU
// partition GPX_RESERVE_CUS, the opt-in pool GPX_TWIN_MASKED -- therefore comes from a
// per-device cache that lives as long as the process: a handle that goes away hands its
// masked streams back (synchronised, idle), the next handle that asks for the same mask takes
// them over, and nothing ever calls hipStreamDestroy on one. The runtime reclaims the queues
// at process exit.
struct MaskedStream {
    hipStream_t s;
    uint32_t mask[32];
    bool busy;
};
static std::mutex g_masked_mu;
static std::vector<MaskedStream> g_masked[64];

static int masked_stream_acquire(int device, int ncu, const uint32_t *mask, hipStream_t *out)
{
    if (device < 0 || device >= 64) {
        gpx_set_error("masked stream: device %d", device);
        return -1;
    }
    std::lock_guard<std::mutex> lock(g_masked_mu);
    for (MaskedStream &m : g_masked[device])
        if (!m.busy && memcmp(m.mask, mask, sizeof m.mask) == 0) {
            m.busy = true;
            *out = m.s;
            return 0;
        }
    MaskedStream m;
    memcpy(m.mask, mask, sizeof m.mask);
    m.busy = true;
    GPX_HIP(hipExtStreamCreateWithCUMask(&m.s, (uint32_t)((ncu + 31) / 32), mask));
    g_masked[device].push_back(m);
    *out = m.s;
    return 0;
}

// back to the cache (true), or not one of the cached streams (false: the caller destroys it)
static bool masked_stream_release(int device, hipStream_t s)
{
    if (device < 0 || device >= 64 || !s) return false;
    std::lock_guard<std::mutex> lock(g_masked_mu);
    for (MaskedStream &m : g_masked[device])
        if (m.s == s) {
            (void)hipStreamSynchronize(s);
            m.busy = false;
            return true;
        }
    return false;
}

static void stream_retire(int device, hipStream_t s)
{
    if (s && !masked_stream_release(device, s)) (void)hipStreamDestroy(s);
}

#define GPX_TWIN_POOL 7
struct TwinPool {
    std::mutex mu;
    bool made = false;
    int users = 0;                 // batch contexts that hold one of the streams
    hipStream_t spacer[GPX_TWIN_POOL] = {};
    hipStream_t stream[GPX_TWIN_POOL] = {};
};
static TwinPool g_twin_pool[64];
static thread_local std::vector<hipStream_t> g_twin_avoid;   // streams earlier in the chain

// do kernels of streams a and b queue behind each other? A kernel that spins for 0.3 ms on
// a, an empty one on b: the empty one ends first unless it had to wait for the spinner.
static bool streams_share_a_queue(hipStream_t a, hipStream_t b)
{
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
        if (ea) (void)hipEventDestroy(ea);
        return false;
    }
    int votes = 0;
    for (int rep = 0; rep < 2; ++rep) {          // twice: a busy device may delay b once
        hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, a, 30000LL);
        (void)hipEventRecord(ea, a);
        hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, b, 0LL);
        (void)hipEventRecord(eb, b);
        (void)hipEventSynchronize(eb);
        if (hipEventQuery(ea) == hipSuccess) ++votes;      // the spinner was over already
        (void)hipEventSynchronize(ea);
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return votes == 2;
}

// How well do kernels of streams a and b run side by side? A dispatch-bound kernel (65 536
// one-wave workgroups that do nothing, 60 us alone) on both at once, time of the pair over
// time of one alone, best of five. tools/probe_queues.hip over ten plain streams of a
// process: 1.5-1.7 for most pairs (the workgroup launch rate is shared), 1.8-1.96 for two
// streams on the same hardware queue (one after the other) -- and 3.5-4.3 for every pair
// between two particular queues of the four: two members there spend their launch-bound
// phases at a quarter of the rate. That third kind is what made value-only batches at
// N = 8192 run at 194-200 or at 240-250 evals/s "depending on the order the process
// created its streams in" (rounds 2 and 3).
static double stream_pair_cost(hipStream_t a, hipStream_t b, double *alone_us)
{
    auto run = [&](hipStream_t x, hipStream_t y) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipStreamSynchronize(x);
            if (y) (void)hipStreamSynchronize(y);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(hold_kernel, dim3(65536), dim3(64), 0, x, 0LL);
            if (y) hipLaunchKernelGGL(hold_kernel, dim3(65536), dim3(64), 0, y, 0LL);
            (void)hipStreamSynchronize(x);
            if (y) (void)hipStreamSynchronize(y);
            const double us = std::chrono::duration<double, std::micro>(
                                  std::chrono::steady_clock::now() - t0).count();
            best = std::min(best, us);
        }
        return best;
    };
    if (*alone_us <= 0.0) *alone_us = run(a, nullptr);
    return run(a, b) / *alone_us;
}

// the pool's streams, once, in one order (caller holds the pool's mutex)
static int twin_pool_make(TwinPool &p, int device, int ncu)
{
    if (p.made) return 0;
    // Plain streams (round 4). Rounds 3-4 made every pool stream a full-mask CU-masked stream,
    // because the runtime gives such a stream a hardware queue of its own (plain streams share
    // four) and a context could then always find one that runs well beside the streams before
    // it (64 value-only thetas at N = 8192 through three contexts: 250 evals/s in every order
    // against 200-215 on a handle that had done single evaluations). Batches of that kind run
    // in groups now (group.hip) and no longer come here, and DESTROYING masked streams turned
    // out not to be safe: hipStreamDestroy of a pool stream did not return in 3 of about 90
    // pool teardowns at the end of round 4 (tests that open and close a handle per case;
    // DESIGN.md section 4), always inside twin_pool_release. GPX_TWIN_MASKED=1 brings the
    // masked queues back (=2: plain streams, each behind an unused masked queue, the
    // arrangement of the first half of round 3); since round 5 they come from the
    // process-lifetime cache above and are never destroyed.
    static const int masked_env = getenv("GPX_TWIN_MASKED") ? atoi(getenv("GPX_TWIN_MASKED")) : 0;
    const bool masked = masked_env > 0 && ncu >= 1 && ncu <= 1024;
    uint32_t mask[32] = {};
    for (int i = 0; masked && i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
    for (int i = 0; i < GPX_TWIN_POOL; ++i) {
        if (masked && masked_env == 1) {
            GPX_TRY(masked_stream_acquire(device, ncu, mask, &p.stream[i]));
            continue;
        }
        if (masked) GPX_TRY(masked_stream_acquire(device, ncu, mask, &p.spacer[i]));
        GPX_HIP(hipStreamCreateWithFlags(&p.stream[i], hipStreamNonBlocking));
    }
    p.made = true;
    return 0;
}

// stream of the index-th (1-based) batch context of a chain on `device` (current device):
// the first pool stream from position index - 1 on that shares no queue with `avoid`
static int twin_pool_stream(int device, int index, int ncu,
                            const std::vector<hipStream_t> &avoid, hipStream_t *out)
{
    if (device < 0 || device >= 64 || index < 1) {
        gpx_set_error("twin stream: device %d index %d", device, index);
        return -1;
    }
    TwinPool &p = g_twin_pool[device];
    std::lock_guard<std::mutex> lock(p.mu);
    GPX_TRY(twin_pool_make(p, device, ncu));
    ++p.users;
    // the first pool stream from position index - 1 on that runs well beside every stream
    // earlier in the chain: not on the same queue, not one of the bad pairs (cost 3.5+
    // against 1.6-1.7; the threshold sits well above what host-side timing adds); failing
    // that, the best of the ones on queues of their own
    static const bool probe = !(getenv("GPX_TWIN_PROBE") && !atoi(getenv("GPX_TWIN_PROBE")));
    int pick = (index - 1) % GPX_TWIN_POOL;
    double best = 1e30, alone = 0.0;
    for (int k = 0; probe && k < GPX_TWIN_POOL; ++k) {
        const int c = (index - 1 + k) % GPX_TWIN_POOL;
        double worst = 0.0;
        for (hipStream_t a : avoid) {
            if (a == p.stream[c] || streams_share_a_queue(a, p.stream[c])) {
                worst = 99.0;
                break;
            }
            worst = std::max(worst, stream_pair_cost(a, p.stream[c], &alone));
        }
        static const bool log = getenv("GPX_TWIN_LOG") != nullptr;
        if (log) fprintf(stderr, "gpx: batch context %d, pool stream %d: cost %.2f\n", index, c, worst);
        if (worst < best) {
            best = worst;
            pick = c;
        }
        if (worst < 2.5) break;
    }
    *out = p.stream[pick];
    return 0;
}

// a batch context goes away; the pool goes with the last one (live CU-masked queues at
// process exit crashed the profiler's teardown) and is rebuilt on the next use
static void twin_pool_release(int device)
{
    if (device < 0 || device >= 64) return;
    TwinPool &p = g_twin_pool[device];
    std::lock_guard<std::mutex> lock(p.mu);
    if (p.users > 0 && --p.users == 0 && p.made) {
        static const bool dlog = getenv("GPX_DESTROY_LOG") != nullptr;
        for (int i = 0; i < GPX_TWIN_POOL; ++i) {
            if (dlog) {
                fprintf(stderr, "twin_pool_release: stream %d\n", i);
                fflush(stderr);
            }
            stream_retire(device, p.stream[i]);        // (masked ones go back to the cache)
            stream_retire(device, p.spacer[i]);
            p.stream[i] = p.spacer[i] = nullptr;
        }
        p.made = false;
    }
}

static bool lookahead_enabled()
{
    static const bool on = !(getenv("GPX_LOOKAHEAD") && !atoi(getenv("GPX_LOOKAHEAD")));
    return on;
}

// streams and events of the look-ahead (chol.hip) for context h on the current device. A
// handle gets them at creation; a batch context when a batch first runs its members with
// look-ahead (gpx_loglik_batch).
static int create_lookahead_streams(gpx_ctx *h)
{
    if (h->crit) return 0;
    hipDeviceProp_t prop;
    GPX_HIP(hipGetDeviceProperties(&prop, h->device));
    int lo = 0, hi = 0;                        // numerically lower = higher priority
    GPX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    GPX_HIP(hipStreamCreateWithPriority(&h->crit, hipStreamNonBlocking, hi));
    // trailing updates (bulk) and the inverse columns (aux) run on every CU but
    // GPX_RESERVE_CUS (default 32, four per XCD; the products are power-bound
    // rather than CU-bound: with 224 CUs they lose nothing measurable): a 128-KB
    // leaf workgroup of the next diagonal
    // block never finds room on a CU that holds two 72-KB GEMM workgroups. Mask
    // bit i is a CU of XCD i % 8 (the driver deals the bits round-robin to the
    // XCDs), so the low bits take the same number of CUs from every XCD.
    // Default 0 since late in round 2: no CU masks. A masked kernel runs at the pace
    // of its CUs (224 of 256: a 7260-workgroup K^-1 update takes 6.93 ms against
    // 6.02), a mask that takes CUs from some shader engines only at the pace of the
    // emptiest one (240 or 248 CUs are no faster than 224), and since the diagonal
    // blocks became one panel launch each they lose less by waiting for the tail of a
    // product launch than the products lose to the mask: one evaluation at
    // N = 16384 75.9 -> 71.6 ms, 2-4 % at N = 2048 ... 8192, value-only 29.2 -> 28.1.
    // GPX_RESERVE_CUS=32 restores the partition (strict above np = 8192).
    static const int reserve = getenv("GPX_RESERVE_CUS") ? atoi(getenv("GPX_RESERVE_CUS")) : 0;
    const int ncu = prop.multiProcessorCount;
    if (reserve > 0 && reserve < ncu && ncu <= 1024) {
        uint32_t mask[32] = {};
        for (int i = reserve; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
        GPX_TRY(masked_stream_acquire(h->device, ncu, mask, &h->bulk));
        GPX_TRY(masked_stream_acquire(h->device, ncu, mask, &h->aux));
        h->bulk_slots = 2 * (ncu - reserve);
        static const int crit_mask = getenv("GPX_CRIT_MASK") ? atoi(getenv("GPX_CRIT_MASK")) : 1;
        if (crit_mask) {
            uint32_t cm[32] = {};
            for (int i = 0; i < reserve; ++i) cm[i / 32] |= 1u << (i % 32);
            GPX_TRY(masked_stream_acquire(h->device, ncu, cm, &h->crit_only));
        }
    } else {
        // A batch context that runs the look-ahead (large batch members) takes its own
        // stream for the trailing updates too: that stream was picked on a queue that runs
        // well beside the other members' (twin_pool_stream), a fresh stream would land on
        // whichever of the four plain queues is least used -- possibly another member's --
        // and build / vector kernels and trailing updates of ONE member have nothing to
        // run side by side for (the updates wait for the build, the vector kernels for
        // them). GPX_TWIN_OWN_BULK=0: a stream of its own, as before.
        static const bool own_bulk = !(getenv("GPX_TWIN_OWN_BULK") && !atoi(getenv("GPX_TWIN_OWN_BULK")));
        if (h->stream_borrowed && own_bulk) {
            h->bulk = h->stream;
            h->bulk_borrowed = true;
        } else {
            GPX_HIP(hipStreamCreateWithFlags(&h->bulk, hipStreamNonBlocking));
        }
        GPX_HIP(hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, lo));
    }
    for (hipEvent_t &e : h->la_events)
        GPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}

int gpx_create(int device, gpx_t **out)
{
    if (!out) {
        gpx_set_error("gpx_create: null out");
        return -1;
    }
    *out = nullptr;
    int ndev = 0;
    GPX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        gpx_set_error("gpx_create: device %d out of range (0..%d)", device, ndev - 1);
        return -1;
    }
    GPX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GPX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gpx_set_error("gpx_create: device %d is %s; libgpx is built for gfx950 only",
                      device, prop.gcnArchName);
        return -1;
    }
    gpx_ctx *h = new (std::nothrow) gpx_ctx();
    if (!h) {
        gpx_set_error("gpx_create: out of host memory");
        return -1;
    }
    h->device = device;
    if (g_creating_twin == 2) {
        // batch context: its stream comes from the device's pool (see above); it never
        // runs the look-ahead, so it needs no other stream
        GPX_TRY(twin_pool_stream(device, g_twin_index, prop.multiProcessorCount, g_twin_avoid,
                                 &h->stream));
        h->stream_borrowed = true;
        h->twin_index = g_twin_index;
        h->chain_streams = g_twin_avoid;
    } else {
        GPX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    }
    for (int i = 0; i <= GPX_NTIMERS; ++i) GPX_HIP(hipEventCreate(&h->ev[i]));
    if (lookahead_enabled() && g_creating_twin != 2) GPX_TRY(create_lookahead_streams(h));
    GPX_TRY(gpx_gemm_init());
    GPX_TRY(gpx_leaf2_init());
    GPX_TRY(gpx_panel_init());
    GPX_TRY(gpx_kmat_init());
    GPX_TRY(h->info.reserve(64));
    GPX_TRY(h->pctl.reserve(gpx_panel_ctl_bytes()));
    // on the handle's own stream and waited for: the streams of a handle do not
    // synchronise with the null stream, and a device memset / device-to-device copy
    // through the synchronous API may return before it has run -- the first panel
    // launch of a freshly created twin context could then meet a control block that
    // is cleared under it (seen once as a wrong lZ with two host threads)
    GPX_HIP(hipMemsetAsync(h->pctl.p, 0, gpx_panel_ctl_bytes(), h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    // [0..2] scalar terms, [3] status word, [4..] trace accumulators: one result copy
    GPX_TRY(h->scalars.reserve((4 + GPX_MAX_HYPER + 2) * sizeof(double)));
    h->acc.p = h->scalars.as<double>() + 4;             // a view, never released on its own
    h->acc.bytes = 0;
    GPX_HIP(hipHostMalloc((void **)&h->hres, (GPX_MAX_HYPER + 8) * sizeof(double)));
    GPX_HIP(hipHostMalloc((void **)&h->hinfo, 64));
    *out = h;
    return 0;
}

int gpx_destroy(gpx_t *h)
{
    if (!h) return 0;
    static const bool dlog = getenv("GPX_DESTROY_LOG") != nullptr;
#define DLOG(msg) do { if (dlog) { fprintf(stderr, "gpx_destroy %p: %s\n", (void *)h, msg); fflush(stderr); } } while (0)
    DLOG("enter");
    if (h->twin) {
        gpx_destroy(h->twin);
        h->twin = nullptr;
    }
    (void)hipSetDevice(h->device);
    if (h->groups) {
        DLOG("groups");
        gpx_groups_destroy(h->groups);
        h->groups = nullptr;
    }
    DLOG("sync stream");
    (void)hipStreamSynchronize(h->stream);
    DLOG("release buffers");
    DevBuf *bufs[] = {&h->X, &h->y, &h->Xf32, &h->A, &h->W, &h->Kinv, &h->r, &h->a,
                      &h->alpha, &h->scalars, &h->partial, &h->info, &h->gv_part, &h->pctl, &h->Ks, &h->KsT,
                      &h->Xs, &h->mu, &h->s2, &h->post_part, &h->t0, &h->t1, &h->t2, &h->split, &h->gpart};
    for (DevBuf *b : bufs) b->release();
    DLOG("events");
    for (int i = 0; i <= GPX_NTIMERS; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    DLOG("host buffers");
    if (h->hres) (void)hipHostFree(h->hres);
    if (h->hinfo) (void)hipHostFree(h->hinfo);
    DLOG("look-ahead events");
    for (hipEvent_t e : h->la_events)
        if (e) (void)hipEventDestroy(e);
    DLOG("stream crit");
    if (h->crit) (void)hipStreamDestroy(h->crit);
    DLOG("stream crit_only");
    stream_retire(h->device, h->crit_only);            // (masked streams: back to the cache,
    DLOG("stream bulk");                               //  never destroyed)
    if (!h->bulk_borrowed) stream_retire(h->device, h->bulk);
    DLOG("stream aux");
    stream_retire(h->device, h->aux);
    DLOG("stream main");
    if (h->stream && !h->stream_borrowed) (void)hipStreamDestroy(h->stream);
    DLOG("pool");
    if (h->stream_borrowed) twin_pool_release(h->device);
    delete h;
    DLOG("done");
#undef DLOG
    return 0;
}

// Safe mode: no task-queue launches on this handle (its batch contexts and groups included).
// Their workgroups hold whole CUs and wait for each other; within one process the launches
// are ordered on the device, but ANOTHER process on the same GPU can starve them until the
// wait bound turns the call into an error ("the panel kernel timed out ..."). The Python
// layer switches a handle to safe mode when that happens and repeats the call: diagonal
// blocks then go by recursion down to the 128-tile leaf (what GPX_PANEL=0 selects for a whole
// process) -- slower below N = 8192, another order of arithmetic (same tolerances, not the
// same bits), no kernel that waits for another workgroup.
int gpx_set_safe_mode(gpx_t *h, int on)
{
    CHECK_H(h);
    for (gpx_ctx *c = h; c; c = c->twin) {
        c->no_panel = on != 0;
        c->have_factor = c->have_inverse = false;
    }
    gpx_groups_safe_mode(&h->groups, h->device, on != 0);
    return 0;
}

int gpx_get_safe_mode(gpx_t *h, int *on)
{
    CHECK_H(h);
    if (!on) {
        gpx_set_error("gpx_get_safe_mode: null output");
        return -1;
    }
    *on = h->no_panel ? 1 : 0;
    return 0;
}

int gpx_synchronize(gpx_t *h)
{
    CHECK_H(h);
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int gpx_enable_timing(gpx_t *h, int on)
{
    CHECK_H(h);
    h->timing = on != 0;
    // (the groups of gpx_loglik_batch keep their own sums: gpx_batch_timings)
    gpx_groups_timing(&h->groups, h->device, h->timing, h->timing);
    return 0;
}

int gpx_batch_timings(gpx_t *h, double *dense_ms, int64_t *members)
{
    CHECK_H(h);
    if (!dense_ms || !members) {
        gpx_set_error("gpx_batch_timings: null outputs");
        return -1;
    }
    gpx_groups_get_timing(h->groups, dense_ms, members);
    return 0;
}

int gpx_get_timings(gpx_t *h, double *ms, int n)
{
    CHECK_H(h);
    for (int i = 0; i < n && i < GPX_NTIMERS; ++i) ms[i] = h->ms[i];
    return 0;
}

const char *gpx_timing_name(int i)
{
    return (i >= 0 && i < GPX_NTIMERS) ? kTimerNames[i] : "";
}

// ---- Kernel.get / Kernel.grad ------------------------------------------------
}  // extern "C"

template <typename T>
static int kernel_get_t(gpx_ctx *h, const KParams &kp, const T *X1, int n1, const T *X2,
                        int n2, int d, T *out)
{
    const bool sym = (X2 == nullptr);
    const int np1 = round_up(n1, 64), np2 = round_up(n2, 64);
    GPX_TRY(h->t0.reserve((size_t)n1 * d * sizeof(T)));
    GPX_TRY(h->t2.reserve((size_t)np1 * np2 * sizeof(T)));
    GPX_HIP(hipMemcpyAsync(h->t0.p, X1, (size_t)n1 * d * sizeof(T), hipMemcpyHostToDevice,
                           h->stream));
    const T *dX2 = h->t0.as<T>();
    if (!sym) {
        GPX_TRY(h->t1.reserve((size_t)n2 * d * sizeof(T)));
        GPX_HIP(hipMemcpyAsync(h->t1.p, X2, (size_t)n2 * d * sizeof(T),
                               hipMemcpyHostToDevice, h->stream));
        dX2 = h->t1.as<T>();
    }
    GPX_TRY(gpx_kbuild<T>(h->stream, kp, h->t0.as<T>(), n1, np1, dX2, n2, np2, d,
                          h->t2.as<T>(), np2, false, false, 0.0));
    GPX_HIP(hipMemcpy2DAsync(out, (size_t)n2 * sizeof(T), h->t2.p, (size_t)np2 * sizeof(T),
                             (size_t)n2 * sizeof(T), n1, hipMemcpyDeviceToHost,
                             h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" {

int gpx_kernel_get(gpx_t *h, const gpx_kspec *k, const void *X1, int64_t n1,
                   const void *X2, int64_t n2, int64_t d, int dtype, void *out)
{
    CHECK_H(h);
    if (!X1 || !out || n1 < 0 || (X2 && n2 < 0)) {
        gpx_set_error("gpx_kernel_get: bad arguments");
        return -1;
    }
    if (!X2) n2 = n1;
    if (n1 == 0 || n2 == 0) return 0;
    if (n1 > (1 << 30) || n2 > (1 << 30)) {
        gpx_set_error("gpx_kernel_get: too many points");
        return -1;
    }
    KParams kp;
    GPX_TRY(gpx_flatten_kspec(k, d, &kp));
    if (dtype == GPX_F64)
        return kernel_get_t<double>(h, kp, (const double *)X1, (int)n1, (const double *)X2,
                                    (int)n2, (int)d, (double *)out);
    if (dtype == GPX_F32)
        return kernel_get_t<float>(h, kp, (const float *)X1, (int)n1, (const float *)X2,
                                   (int)n2, (int)d, (float *)out);
    gpx_set_error("gpx_kernel_get: unknown dtype %d", dtype);
    return -1;
}

int gpx_kernel_grad(gpx_t *h, const gpx_kspec *k, const double *X1, int64_t n1,
                    const double *X2, int64_t n2, int64_t d, double *out)
{
    CHECK_H(h);
    if (!X1 || !out || n1 < 0 || (X2 && n2 < 0)) {
        gpx_set_error("gpx_kernel_grad: bad arguments");
        return -1;
    }
    if (!X2) n2 = n1;
    if (n1 == 0 || n2 == 0) return 0;
    KParams kp;
    GPX_TRY(gpx_flatten_kspec(k, d, &kp));
    const size_t xb1 = (size_t)n1 * d * 8, xb2 = (size_t)n2 * d * 8;
    const size_t ob = (size_t)kp.nhyper * n1 * n2 * 8;
    GPX_TRY(h->t0.reserve(xb1));
    GPX_TRY(h->t2.reserve(ob));
    GPX_HIP(hipMemcpyAsync(h->t0.p, X1, xb1, hipMemcpyHostToDevice, h->stream));
    const double *dX2 = h->t0.as<double>();
    if (X2) {
        GPX_TRY(h->t1.reserve(xb2));
        GPX_HIP(hipMemcpyAsync(h->t1.p, X2, xb2, hipMemcpyHostToDevice, h->stream));
        dX2 = h->t1.as<double>();
    }
    GPX_TRY(gpx_kgrad(h->stream, kp, h->t0.as<double>(), (int)n1, dX2, (int)n2, (int)d,
                      h->t2.as<double>()));
    GPX_HIP(hipMemcpyAsync(out, h->t2.p, ob, hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

__global__ void f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out,
                                  size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

int gpx_kernel_build_resident(gpx_t *h, const gpx_kspec *k, int dtype, int reps,
                              double *ms)
{
    CHECK_H(h);
    if (h->n <= 0) {
        gpx_set_error("gpx_kernel_build_resident: no data (gpx_set_data first)");
        return -1;
    }
    if (reps < 1) reps = 1;
    KParams kp;
    GPX_TRY(gpx_flatten_kspec(k, h->d, &kp));
    const int n = h->n, np = round_up(n, 64);
    const size_t esz = dtype == GPX_F32 ? 4 : 8;
    GPX_TRY(h->t2.reserve((size_t)np * np * esz));
    if (dtype == GPX_F32) {
        const size_t cnt = (size_t)n * h->d;
        GPX_TRY(h->Xf32.reserve(cnt * 4));
        hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((cnt + 255) / 256)),
                           dim3(256), 0, h->stream, h->X.as<double>(), h->Xf32.as<float>(),
                           cnt);
    }
    hipEvent_t e0 = h->ev[GPX_NTIMERS], e1 = h->ev[0];
    for (int it = -1; it < reps; ++it) {        // one untimed warm-up
        if (it == 0) GPX_HIP(hipEventRecord(e0, h->stream));
        if (dtype == GPX_F32)
            GPX_TRY(gpx_kbuild<float>(h->stream, kp, h->Xf32.as<float>(), n, np,
                                      h->Xf32.as<float>(), n, np, h->d, h->t2.as<float>(),
                                      np, false, false, 0.0));
        else
            GPX_TRY(gpx_kbuild<double>(h->stream, kp, h->X.as<double>(), n, np,
                                       h->X.as<double>(), n, np, h->d, h->t2.as<double>(),
                                       np, false, false, 0.0));
    }
    GPX_HIP(hipEventRecord(e1, h->stream));
    GPX_HIP(hipEventSynchronize(e1));
    float t = 0;
    GPX_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / reps;
    return 0;
}

// ---- ExactGP -----------------------------------------------------------------
int gpx_set_data(gpx_t *h, const double *X, int64_t n, int64_t d, const double *y)
{
    CHECK_H(h);
    if (!X || !y || n < 1 || d < 1 || d > GPX_MAX_DIM || n > (1 << 20)) {
        gpx_set_error("gpx_set_data: bad shape n=%lld d=%lld (d <= %d)", (long long)n,
                      (long long)d, GPX_MAX_DIM);
        return -1;
    }
    // capacity: at least 256 rows of slack, rounded to 1024, so that appended
    // observations (gpx_exact_append) open new 128-blocks without a reallocation;
    // the identity padding makes the slack free numerically, and only np x np is
    // ever touched by a factorisation
    const size_t cap = (size_t)round_up(n + 256, 1024);
    GPX_TRY(h->X.reserve(cap * d * 8));
    GPX_TRY(h->y.reserve(cap * 8));
    GPX_HIP(hipMemcpyAsync(h->X.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, h->stream));
    GPX_HIP(hipMemcpyAsync(h->y.p, y, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    h->n = (int)n;
    h->d = (int)d;
    h->np = round_up(n, GPX_TILE);
    h->cap = (int)cap;
    h->ld = ld_for(h->cap);
    h->data_version++;
    h->have_factor = h->have_inverse = false;
    return 0;
}

static int reserve_factor(gpx_ctx *h, bool inverse)
{
    if (h->cap < h->np) h->cap = h->np;
    const size_t mat = (size_t)h->cap * h->ld * 8, vec = (size_t)h->cap * 8;
    GPX_TRY(h->A.reserve(mat));
    GPX_TRY(h->W.reserve(mat));
    GPX_TRY(h->r.reserve(vec));
    GPX_TRY(h->a.reserve(vec));
    GPX_TRY(h->gv_part.reserve(gpx_trsv_scratch(h->cap) * 8));
    GPX_TRY(h->Kinv.reserve(mat));          // also the scratch of potrf
    if (inverse) {
        GPX_TRY(h->alpha.reserve(vec));
        GPX_TRY(h->partial.reserve(gpx_trace_scratch(h->cap) * 8));
    }
    return 0;
}

// enqueue K build + Cholesky + a; no host sync. grad_follows: enqueue_grad comes next on
// this stream (fused evaluation): the last K^-1 update may still be running on the
// look-ahead's third stream when this returns, enqueue_grad joins it.
static int enqueue_update(gpx_ctx *h, StageClock &clk, int mode, bool grad_follows = false)
{
    DenseWs w = h->ws();
    if (h->kinv_pending) {
        // a deferred K^-1 update that no enqueue_grad joined (its evaluation failed on the
        // way): this build writes the same matrix, so wait for it first
        DenseWs wj = w;
        wj.defer_kinv = true;
        GPX_TRY(gpx_potrf_join(h->stream, wj));
        h->kinv_pending = false;
    }
    const double sn2 = exp(h->log_sn * 2);               // gaussian.py:36-39
    GPX_HIP(hipMemsetAsync(h->info.p, 0, sizeof(int), h->stream));
    // diagonal 128-tiles into A, the others straight into the staging area (Kinv)
    // from which the factorisation's panel products read them. With look-ahead the
    // rows of the first diagonal block are built first and that block is factored
    // while the rest of the matrix is still being built.
    static const int lead_on = getenv("GPX_LEAD_BUILD") ? atoi(getenv("GPX_LEAD_BUILD")) : 1;
    // (two blocks when the first launch may be a wide panel, chol.hip)
    const GpxBlocks lb(h->np, mode == GPX_POTRF_KINV);
    const int lead_rows = lb.count >= 2 && lb.len(1) <= GPX_PANEL_MAX ? lb.off(2) : lb.len(0);
    hipEvent_t lead_ev = gpx_potrf_lead_event(w);
    const bool lead = lead_on && lead_ev && w.crit && lead_rows < h->np;
    if (lead) {
        GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                                   h->X.as<double>(), h->n, h->np, h->d, w.A, h->ld, true,
                                   true, sn2, w.Kinv, 0, lead_rows));
        GPX_HIP(hipEventRecord(lead_ev, h->stream));
        // test hook (tests/test_gpu_la.py): hold the rest of the build back, so that an
        // ordering bug between it and what follows the first diagonal block shows every
        // time instead of once in a busy batch
        static const int hold_us = getenv("GPX_TEST_HOLD_BUILD_US")
                                       ? atoi(getenv("GPX_TEST_HOLD_BUILD_US")) : 0;
        if (hold_us > 0)
            hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, h->stream,
                               (long long)hold_us * 100);
        GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                                   h->X.as<double>(), h->n, h->np, h->d, w.A, h->ld, true,
                                   true, sn2, w.Kinv, lead_rows, h->np - lead_rows));
        w.lead = lead_ev;
        w.lead_rows = lead_rows;
    } else {
        GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                                   h->X.as<double>(), h->n, h->np, h->d, w.A, h->ld, true,
                                   true, sn2, w.Kinv));
    }
    clk.tick(T_BUILD);
    // with the gradient in view, R^-1 and (R^T R)^-1 are built beside the factorisation
    static const int defer_on = getenv("GPX_DEFER_KINV") ? atoi(getenv("GPX_DEFER_KINV")) : 1;
    w.defer_kinv = grad_follows && defer_on && mode == GPX_POTRF_KINV;
    h->kinv_pending = w.defer_kinv;
    h->lz_enqueued = false;              // (a failed evaluation may have left it set)
    // A factorisation that runs as one launch over the whole matrix takes the residual
    // along as one more tile column (column np of the staging matrix, in the padding) and
    // returns a = R^-T r in the same place of A: the forward substitution as tasks of the
    // launch instead of 14 launches behind it (0.13 of 1.77 ms at N = 4096).
    // (round 4: also for the matrices of 256 .. 1024 rows that are one panel, in every mode,
    // so that a comes out of the same arithmetic with or without gradients there)
    // decided here, once: gpx_potrf takes the whole-matrix launch iff w.whole says so
    w.whole = gpx_potrf_whole(w, mode);
    w.full_w = grad_follows && gpx_grad_full_w(w, mode);
    const bool aug = gpx_potrf_rhs_ok(w, mode);
    if (aug) {
        // (one kernel: the residual into column np of the staging matrix, zeros right of it)
        GPX_TRY(gpx_residual_rhs(h->stream, h->y.as<double>(), h->mean, h->n, h->np, nullptr,
                                 w.Kinv, h->ld));
        w.aug_rhs = true;
    }
    GPX_TRY(gpx_potrf(h->stream, w, mode, true));
    const bool full_inverse = mode != GPX_POTRF_R || GpxBlocks(h->np).count == 1 || w.full_w;
    h->w_complete = full_inverse;
    h->kinv_ready = mode == GPX_POTRF_KINV;
    h->posterior_calls = 0;
    // (deferred: the stage ends where K^-1 is complete, in enqueue_grad)
    if (!w.defer_kinv) clk.tick(T_POTRF);
    if (aug) {
        GPX_TRY(gpx_column_out(h->stream, w.A, h->ld, h->np, h->np, h->a.as<double>(),
                               MemberBatch()));
    } else {
        GPX_TRY(gpx_residual(h->stream, h->y.as<double>(), h->mean, h->n, h->np,
                             h->r.as<double>()));
        GPX_TRY(gpx_trsv_rt(h->stream, w, full_inverse, h->r.as<double>(), h->a.as<double>(),
                            h->gv_part.as<double>()));
    }
    if (!w.defer_kinv) clk.tick(T_TRSV);
    return 0;
}

// enqueue K^-1, alpha and the trace terms; no host sync
static int enqueue_grad(gpx_ctx *h, StageClock &clk)
{
    DenseWs w = h->ws();
    if (!h->w_complete) {
        GPX_TRY(gpx_trtri(h->stream, w));
        h->w_complete = true;
        clk.tick(T_TRTRI);
    }
    GPX_TRY(gpx_trmv_upper(h->stream, w.W, h->ld, h->np, h->a.as<double>(),
                           h->alpha.as<double>()));
    if (h->kinv_pending) {
        // a = R^-T r and alpha = R^-1 a ran beside the last K^-1 update of the sweep, and so
        // do the scalar terms (one workgroup, 44 us): wait for the update here; the potrf
        // stage (factor + inverse) ends with it
        GPX_TRY(gpx_lz_terms(h->stream, h->A.as<double>(), h->ld, h->n, h->a.as<double>(),
                             h->alpha.as<double>(), h->scalars.as<double>(), 1, 0, 0, 0,
                             h->info.as<int>()));
        h->lz_enqueued = true;
        w.defer_kinv = true;
        GPX_TRY(gpx_potrf_join(h->stream, w));
        h->kinv_pending = false;
        clk.tick(T_POTRF);
    } else {
        clk.tick(T_TRMV);
    }
    if (!h->kinv_ready) {
        GPX_TRY(gpx_lauum(h->stream, w));
        h->kinv_ready = true;
        clk.tick(T_LAUUM);
    }
    GPX_TRY(gpx_trace_grad(h->stream, h->kp, h->X.as<double>(), h->n, h->np, h->d, w.Kinv,
                           h->ld, h->alpha.as<double>(), h->partial.as<double>(),
                           h->acc.as<double>()));
    clk.tick(T_TRACE);
    return 0;
}

// scalars + asynchronous copy of the few result doubles to pinned host memory
static int enqueue_finish(gpx_ctx *h, StageClock &clk, bool grad)
{
    if (!(grad && h->lz_enqueued))
        GPX_TRY(gpx_lz_terms(h->stream, h->A.as<double>(), h->ld, h->n, h->a.as<double>(),
                             grad ? h->alpha.as<double>() : nullptr, h->scalars.as<double>(), 1,
                             0, 0, 0, h->info.as<int>()));
    h->lz_enqueued = false;
    // scalar terms, status word and (with gradients) the trace accumulators behind them:
    // one copy (three until the end of round 4: 5 us of queue time each)
    GPX_HIP(hipMemcpyAsync(h->hres, h->scalars.p,
                           (4 + (grad ? 1 + h->kp.nhyper : 0)) * sizeof(double),
                           hipMemcpyDeviceToHost, h->stream));
    h->pending_grad = grad;
    clk.tick(T_SCALARS);
    return 0;
}

// the one host sync of an evaluation
static int collect(gpx_ctx *h, StageClock &clk, double *lZ, double *dlZ, int *info)
{
    GPX_HIP(hipStreamSynchronize(h->stream));
    clk.collect();
    const bool grad = h->pending_grad;
    const double *sc = h->hres, *acc = h->hres + 4;
    int inf = (int)h->hres[3];      // (the status word, through lz_terms_kernel)
    if (info) *info = inf;
    if (inf < 0) {                  // panel kernel gave up waiting (panel.hip)
        h->have_factor = h->have_inverse = false;
        gpx_set_error("internal: the panel kernel timed out waiting for a dependency");
        return -1;
    }
    if (inf > h->n) inf = 0;        // cannot happen: the padding is the identity
    if (inf != 0) {
        h->have_factor = h->have_inverse = false;
        gpx_set_error("matrix is not positive definite: pivot %d", inf);
        return inf;
    }
    // exact.py:119-121
    h->lZ = gpx_assemble_lz(sc, h->n);
    if (lZ) *lZ = h->lZ;
    if (grad && dlZ) gpx_assemble_dlz(sc, acc, exp(h->log_sn * 2), h->kp.nhyper, dlZ);
    return 0;
}

static int finish(gpx_ctx *h, StageClock &clk, bool grad, double *lZ, double *dlZ,
                  int *info)
{
    GPX_TRY(enqueue_finish(h, clk, grad));
    return collect(h, clk, lZ, dlZ, info);
}

static int check_ready(gpx_ctx *h, const gpx_kspec *k, double log_sn, double mean)
{
    if (h->n <= 0) {
        gpx_set_error("no data: call gpx_set_data first");
        return -1;
    }
    if (!std::isfinite(log_sn) || !std::isfinite(mean)) {
        gpx_set_error("non-finite hyperparameters");
        return -1;
    }
    GPX_TRY(gpx_flatten_kspec(k, h->d, &h->kp));
    h->log_sn = log_sn;
    h->mean = mean;
    return 0;
}

int gpx_exact_update(gpx_t *h, const gpx_kspec *k, double log_sn, double mean, int *info)
{
    CHECK_H(h);
    GPX_TRY(check_ready(h, k, log_sn, mean));
    GPX_TRY(reserve_factor(h, false));
    h->have_factor = h->have_inverse = false;
    StageClock clk(h);
    GPX_TRY(enqueue_update(h, clk, GPX_POTRF_R));
    int r = finish(h, clk, false, nullptr, nullptr, info);
    if (r == 0) h->have_factor = true;
    return r;
}

int gpx_exact_loglik(gpx_t *h, double *lZ, double *dlZ)
{
    CHECK_H(h);
    if (!h->have_factor) {
        gpx_set_error("gpx_exact_loglik: no factorisation (call gpx_exact_update)");
        return -1;
    }
    if (!dlZ) {
        if (lZ) *lZ = h->lZ;
        return 0;
    }
    GPX_TRY(reserve_factor(h, true));
    StageClock clk(h);
    GPX_TRY(enqueue_grad(h, clk));
    int r = finish(h, clk, true, lZ, dlZ, nullptr);
    if (r == 0) h->have_inverse = true;
    return r;
}

int gpx_exact_eval(gpx_t *h, const gpx_kspec *k, double log_sn, double mean,
                   int want_grad, double *lZ, double *dlZ, int *info)
{
    CHECK_H(h);
    GPX_TRY(check_ready(h, k, log_sn, mean));
    const bool grad = want_grad && dlZ;
    GPX_TRY(reserve_factor(h, grad));
    h->have_factor = h->have_inverse = false;
    StageClock clk(h);
    GPX_TRY(enqueue_update(h, clk, grad ? gpx_grad_mode(h->ws()) : GPX_POTRF_R, grad));
    if (grad) GPX_TRY(enqueue_grad(h, clk));
    int r = finish(h, clk, grad, lZ, dlZ, info);
    if (r == 0) {
        h->have_factor = true;
        h->have_inverse = grad;
    }
    return r;
}

// enqueue one whole evaluation on h's stream without waiting for it
static int eval_enqueue(gpx_ctx *h, const gpx_kspec *k, double log_sn, double mean,
                        bool grad, StageClock &clk)
{
    GPX_HIP(hipSetDevice(h->device));
    GPX_TRY(check_ready(h, k, log_sn, mean));
    GPX_TRY(reserve_factor(h, grad));
    h->have_factor = h->have_inverse = false;
    const int mode = grad ? gpx_grad_mode(h->ws()) : GPX_POTRF_R;
    GPX_TRY(enqueue_update(h, clk, mode, grad));
    if (grad) GPX_TRY(enqueue_grad(h, clk));
    return enqueue_finish(h, clk, grad);
}

static int ensure_twin(gpx_ctx *h)
{
    if (!h->twin) {
        g_creating_twin = 2;
        g_twin_index = h->twin_index + 1;
        g_twin_avoid = h->chain_streams;               // every stream before it in the chain
        g_twin_avoid.push_back(h->stream);
        if (h->bulk && h->bulk != h->stream) g_twin_avoid.push_back(h->bulk);   // (the handle's)
        const int rc = gpx_create(h->device, &h->twin);
        g_creating_twin = 0;
        GPX_TRY(rc);
        h->twin->no_panel = h->no_panel;
    }
    gpx_ctx *t = h->twin;
    if (t->n != h->n || t->d != h->d || t->data_version != h->data_version) {
        GPX_TRY(t->X.reserve((size_t)h->n * h->d * 8));
        GPX_TRY(t->y.reserve((size_t)h->n * 8));
        // on the twin's stream: its evaluations are ordered behind the copy
        GPX_HIP(hipMemcpyAsync(t->X.p, h->X.p, (size_t)h->n * h->d * 8,
                               hipMemcpyDeviceToDevice, t->stream));
        GPX_HIP(hipMemcpyAsync(t->y.p, h->y.p, (size_t)h->n * 8, hipMemcpyDeviceToDevice,
                               t->stream));
        GPX_HIP(hipStreamSynchronize(t->stream));
        t->n = h->n; t->d = h->d; t->np = h->np; t->ld = h->ld; t->cap = h->cap;
        t->data_version = h->data_version;
        t->have_factor = t->have_inverse = false;
    }
    return 0;
}

// ---- ExactGP._updateinc (exact.py:57-62) --------------------------------------
// One 128-block column [j0, j0 + 128) of the factor is (re)built for the points that
// were appended into it. With W = R^-1 complete on the leading j0 x j0 part (an
// invariant appends keep once it holds) every step is one triangle-aware product of
// O(j0^2 128) flops -- no substitution, no recursion over a block tree:
//   B    = K(X[:j0], X[j0:j0+128])                      strip build (staging: Kinv)
//   X    = W_top^T B                                   -> R[:j0, j0:]   (into A)
//   S    = K(X[j0:], X[j0:]) + sn2 I - X^T X           Schur complement (split-k)
//   R_nn, W_nn = leaf(S)                               128 x 128 Cholesky + inverse
//   W[:j0, j0:] = -W_top (X W_nn)                      keeps W complete
static int append_block(gpx_ctx *h, int j0)
{
    const DenseWs w = h->ws();
    const int ld = h->ld;
    const double sn2 = exp(h->log_sn * 2);
    hipStream_t s = h->stream;
    double *Bst = w.Kinv + j0;                         // staging strip, rows [0, j0)
    double *Xst = w.A + j0;                            // R[:j0, j0:j0+128]
    double *Sdiag = w.A + (size_t)j0 * ld + j0;
    double *Wdiag = w.W + (size_t)j0 * ld + j0;
    GPX_TRY(gpx_kbuild_strip(s, h->kp, h->X.as<double>(), h->n, h->np, j0, GPX_TILE, h->d,
                             w.A, ld, sn2, w.Kinv));
    auto product = [&](int ta, const double *A, const double *B, double *C, int M, int K,
                       double alpha, int flags) {
        GemmArgs g;
        g.A = A; g.B = B; g.C = C;
        g.lda = g.ldb = g.ldc = ld;
        g.M = M; g.N = GPX_TILE; g.K = K;
        g.alpha = alpha; g.beta = 0.0;
        g.strideA = g.strideB = g.strideC = 0;
        g.batch = 1;
        g.flags = flags;
        g.tile = 64; g.order = 0; g.swizzle = 0; g.waves = 0; g.use_lists = 1;
        g.tiles = nullptr;
        return gpx_gemm(s, ta, 0, g);
    };
    if (j0 > 0) {
        // X = W_top^T B: op(A)[m][k] = W[k][m] is lower triangular, k < m0 + tile
        GPX_TRY(product(1, w.W, Bst, Xst, j0, j0, 1.0, GEMM_KHI_M));
        // S -= X^T X: one 128 x 128 tile with k = j0, cut into k-chunks that run side
        // by side; their partial products are added in a fixed order
        const int kc = 512;
        const int nsplit = (j0 + kc - 1) / kc;
        const long long stride = (long long)GPX_TILE * GPX_TILE;
        GPX_TRY(h->split.reserve((size_t)nsplit * stride * 8));
        GemmArgs g;
        g.A = Xst; g.B = Xst; g.C = h->split.as<double>();
        g.lda = g.ldb = ld; g.ldc = GPX_TILE;
        g.M = g.N = GPX_TILE; g.K = j0;
        g.alpha = 1.0; g.beta = 0.0;
        g.strideA = g.strideB = 0; g.strideC = stride;
        g.batch = nsplit; g.kchunk = kc;
        g.flags = 0;
        g.tile = 64; g.order = 0; g.swizzle = 0; g.waves = 0; g.use_lists = 1;
        g.tiles = nullptr;
        GPX_TRY(gpx_gemm(s, 1, 0, g));
        GPX_TRY(gpx_sub_partials(s, h->split.as<double>(), nsplit, stride, GPX_TILE, GPX_TILE,
                                 Sdiag, ld));
    }
    GPX_TRY(gpx_potrf_leaf2(s, Sdiag, ld, Wdiag, ld, w.info, j0));
    if (j0 > 0) {
        // T = X W_nn (staging strip), then W[:j0, j0:] = -W_top T (W upper: k >= m0)
        GPX_TRY(product(0, Xst, Wdiag, Bst, j0, GPX_TILE, 1.0, 0));
        GPX_TRY(product(0, w.W, Bst, w.W + j0, j0, j0, -1.0, GEMM_KLO_M));
    }
    return 0;
}

// ExactGP._updateinc (exact.py:57-62): m new observations appended to the data of
// the current factorisation in O(n^2 m). The handle reserves capacity beyond n
// (gpx_set_data), so new 128-blocks open in place; the first append after a
// value-only factorisation completes R^-1 once (a fraction of a factorisation),
// every later one keeps it complete. Returns -3 (no error text) when no
// factorisation is current or the capacity is exhausted: the caller then
// refactorises, like the reference's NotImplementedError fallback (_base.py:132-141).
// If the extended matrix is not positive definite the handle is rolled back to the
// old point count and needs a new gpx_exact_update.
int gpx_exact_append(gpx_t *h, const double *Xnew, const double *ynew, int64_t m, int *info)
{
    CHECK_H(h);
    if (!Xnew || !ynew || m < 1) {
        gpx_set_error("gpx_exact_append: bad arguments");
        return -1;
    }
    if (!h->have_factor || h->n <= 0 || h->n + m > h->cap) return -3;
    const int n_old = h->n, np_old = h->np;
    GPX_HIP(hipMemcpyAsync(h->X.as<double>() + (size_t)n_old * h->d, Xnew,
                           (size_t)m * h->d * 8, hipMemcpyHostToDevice, h->stream));
    GPX_HIP(hipMemcpyAsync(h->y.as<double>() + n_old, ynew, (size_t)m * 8,
                           hipMemcpyHostToDevice, h->stream));
    StageClock clk(h);
    GPX_HIP(hipMemsetAsync(h->info.p, 0, sizeof(int), h->stream));
    if (!h->w_complete) {
        GPX_TRY(gpx_trtri(h->stream, h->ws()));
        h->w_complete = true;
        clk.tick(T_TRTRI);
    }
    h->data_version++;
    h->have_factor = h->have_inverse = false;
    h->kinv_ready = false;
    h->posterior_calls = 0;
    int rc = 0;
    for (int64_t done = 0; done < m && rc == 0;) {
        const int j0 = h->n / GPX_TILE * GPX_TILE;     // block column that takes the next points
        const int take = (int)std::min<int64_t>(m - done, j0 + GPX_TILE - h->n);
        h->n += take;
        h->np = j0 + GPX_TILE;
        rc = append_block(h, j0);
        done += take;
    }
    clk.tick(T_POTRF);
    if (rc == 0) {
        rc = gpx_residual(h->stream, h->y.as<double>(), h->mean, h->n, h->np,
                          h->r.as<double>());
        if (rc == 0)
            rc = gpx_trsv_rt(h->stream, h->ws(), true, h->r.as<double>(), h->a.as<double>(),
                             h->gv_part.as<double>());
        clk.tick(T_TRSV);
    }
    if (rc == 0) rc = finish(h, clk, false, nullptr, nullptr, info);
    if (rc == 0) {
        h->have_factor = true;
    } else {
        (void)hipStreamSynchronize(h->stream);
        h->n = n_old;                                  // the old data stand; R is gone
        h->np = np_old;
        h->w_complete = false;
    }
    return rc;
}

// does gpx_loglik_batch / gpx_posterior_batch hand a batch of B to the groups at all?
static bool batch_in_groups(const gpx_ctx *h, int64_t B)
{
    // (one tile and up: the reference's own demo sizes, N = 5 ... 128, are a batched leaf)
    return B >= 2 && h->np <= gpx_groups_max_np() && h->np >= GPX_TILE &&
           (h->np <= 8192 || B >= gpx_groups_min_big());
}

int gpx_batch_plan(gpx_t *h, int64_t B, int want_grad, int *plan)
{
    CHECK_H(h);
    if (!plan || B < 0) {
        gpx_set_error("gpx_batch_plan: bad arguments");
        return -1;
    }
    if (h->n <= 0) {
        gpx_set_error("no data: call gpx_set_data first");
        return -1;
    }
    plan[0] = plan[1] = plan[2] = plan[3] = 0;
    int members = 0, inflight = 0, lockstep = 0;
    if (batch_in_groups(h, B))
        GPX_TRY(gpx_groups_plan(h->groups, h->np, B, want_grad != 0, &members, &inflight, &lockstep));
    plan[3] = h->no_panel ? 1 : 0;                     // safe mode (gpx_set_safe_mode)
    if (members > 0) {
        plan[0] = lockstep == 2 ? 3 : (lockstep ? 2 : 1);
        plan[1] = members;
        plan[2] = inflight;
    } else {
        // one context and stream per member (rounds 1-3), up to three in flight
        plan[1] = 1;
        plan[2] = (int)std::min<int64_t>(3, std::max<int64_t>(B, 1));
    }
    return 0;
}

int gpx_loglik_batch(gpx_t *h, const gpx_kspec *k, const double *thetas, int64_t B,
                     int want_grad, double *lZ, double *dlZ, int *info)
{
    CHECK_H(h);
    if (!k || !thetas || !lZ || B < 0) {
        gpx_set_error("gpx_loglik_batch: bad arguments");
        return -1;
    }
    if (h->n <= 0) {
        gpx_set_error("no data: call gpx_set_data first");
        return -1;
    }
    const int nth = 1 + k->nhyper + 1;
    const bool grad = want_grad && dlZ;
    // Up to np = 32768: groups of members in lock-step, every kernel one launch over the whole
    // group (group.hip; round 4). A member takes the arithmetic of the same evaluation on
    // its own, whichever of the two paths runs it.
    if (batch_in_groups(h, B)) {
        const int rc = gpx_groups_loglik(&h->groups, h->device, h->X.as<double>(),
                                         h->y.as<double>(), h->n, h->d, h->np, k, thetas, B, grad,
                                         lZ, dlZ, info);
        if (rc != 1) {                                 // (1: declined, the contexts below run it)
            h->have_factor = h->have_inverse = false;  // (as below: the handle held a member)
            return rc;
        }
    }
    // Independent evaluations: keep a few in flight (own stream + workspace each) so
    // that the latency-bound diagonal-block chain of one overlaps the MFMA-bound
    // trailing updates of the other. GPX_BATCH_INFLIGHT=1 restores one at a time.
    static const int inflight = [] {
        const char *e = getenv("GPX_BATCH_INFLIGHT");
        const int v = e ? atoi(e) : 3;    // measured best at N = 4096 .. 16384
        return v < 1 || v > 8 ? 3 : v;
    }();
    int depth = (int)std::min<int64_t>(B > 1 ? inflight : 1, B > 0 ? B : 1);
    // every context holds three np x ld matrices: stay well inside 288 GB of HBM
    const double ws_bytes = 3.0 * h->np * (double)h->ld * 8;
    while (depth > 1 && depth * ws_bytes > 160e9) --depth;
    gpx_ctx *ctx[8] = {h, h, h, h, h, h, h, h};
    for (int i = 1; i < depth; ++i) {          // contexts form a chain of twins
        GPX_TRY(ensure_twin(ctx[i - 1]));
        ctx[i] = ctx[i - 1]->twin;
    }
    // Large members also run the look-ahead of the single evaluation, each on streams of
    // its own (round 3): their chain of diagonal blocks hides under their own products
    // AND the members fill each other's bubbles. 9 thetas, 3 in flight: N = 16384 14.04 ->
    // 14.80 evals/s with gradients, 38.1 -> 38.9 value only; N = 12288 32.9 -> 34.1 /
    // 82.8 -> 81.4; N = 10240 54.8 -> 55.3 / 137 -> 128; N = 8192 (64 thetas) 103 -> 104 /
    // 251 -> 217: on from np = 12288 with gradients, from 16384 without
    // (GPX_BATCH_LOOKAHEAD=0 / 1 forces). Four members in flight this way collapse.
    static const int batch_la_env = getenv("GPX_BATCH_LOOKAHEAD") ? atoi(getenv("GPX_BATCH_LOOKAHEAD")) : -1;
    const bool batch_la = depth > 1 && depth <= 3 && lookahead_enabled() &&
                          (batch_la_env >= 0 ? batch_la_env != 0
                                             : (h->np >= 16384 || (grad && h->np >= 12288)));
    if (batch_la)
        for (int i = 0; i < depth; ++i) {
            GPX_HIP(hipSetDevice(h->device));
            GPX_TRY(create_lookahead_streams(ctx[i]));
        }
    for (int i = 0; i < depth; ++i) {
        ctx[i]->batch_la = batch_la;
        ctx[i]->in_batch = depth > 1;
    }
    const bool timing = h->timing;
    h->timing = false;                 // stage events are per single evaluation
    std::vector<gpx_kspec> store[8];
    int rc = 0;
    if (depth > 1) gpx_gemm_concurrency(h->device, +1);
    auto harvest = [&](int64_t b) -> int {
        gpx_ctx *c = ctx[b % depth];
        StageClock clk(c);
        int inf = 0;
        int r = collect(c, clk, &lZ[b], grad ? dlZ + b * nth : nullptr, &inf);
        if (info) info[b] = inf;
        if (r < 0) return r;
        if (r > 0) {                 // not PD: like -inf log-likelihood for a sampler
            lZ[b] = -INFINITY;
            if (grad)
                for (int i = 0; i < nth; ++i) dlZ[b * nth + i] = NAN;
        } else {
            c->have_factor = true;
            c->have_inverse = grad;
        }
        return 0;
    };
    for (int64_t b = 0; b < B && rc >= 0; ++b) {
        if (b >= depth) rc = harvest(b - depth);
        if (rc < 0) break;
        gpx_ctx *c = ctx[b % depth];
        const double *th = thetas + b * nth;
        gpx_kspec kb;
        rc = gpx_kspec_with_hyper(k, th + 1, store[b % depth], &kb);
        if (rc < 0) break;
        StageClock clk(c);
        rc = eval_enqueue(c, &kb, th[0], th[nth - 1], grad, clk);
    }
    for (int64_t b = std::max<int64_t>(0, B - depth); b < B && rc >= 0; ++b) rc = harvest(b);
    if (depth > 1) gpx_gemm_concurrency(h->device, -1);
    for (int i = 0; i < depth; ++i) ctx[i]->batch_la = ctx[i]->in_batch = false;
    (void)hipSetDevice(h->device);
    h->timing = timing;
    // the caller's context ran batch members too: whatever update it held before is
    // gone, and the last member's factor is not something a later gpx_exact_loglik /
    // posterior / append on this handle may silently pick up
    h->have_factor = h->have_inverse = false;
    return rc < 0 ? rc : 0;
}

// C (np x mcp) = op(W) B for the triangular W = R^-1 (ta = 1: W^T, k < m0 + tile;
// ta = 0: W, k >= m0). With few columns the launch has a handful of tiles whose k
// ranges reach np and is as long as its longest tile: then k is cut into chunks
// that run as separate workgroups and the partial products are summed in a fixed
// order (deterministic).
static int tri_product(gpx_ctx *h, int ta, const double *B, double *C, int mcp)
{
    const DenseWs w = h->ws();
    GemmArgs g;
    g.A = w.W; g.B = B; g.C = C;
    g.lda = h->ld; g.ldb = mcp; g.ldc = mcp;
    g.M = h->np; g.N = mcp; g.K = h->np;
    g.alpha = 1.0; g.beta = 0.0;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = ta ? GEMM_KHI_M : GEMM_KLO_M;
    g.tile = 0; g.order = ta ? 1 : 0; g.swizzle = 0; g.waves = 0; g.use_lists = 1;
    g.tiles = nullptr;
    const long long tiles64 = (long long)(h->np / 64) * (mcp / 64);
    if (tiles64 > 1024 || h->np < 2048) return gpx_gemm(h->stream, ta, 0, g);
    const int kc = h->np >= 8192 ? 2048 : 1024;
    const int nsplit = (h->np + kc - 1) / kc;
    const long long stride = (long long)h->np * mcp;
    GPX_TRY(h->split.reserve((size_t)nsplit * stride * 8));
    g.C = h->split.as<double>();
    g.kchunk = kc;
    g.batch = nsplit;
    g.strideC = stride;
    g.tile = 64;
    GPX_TRY(gpx_gemm(h->stream, ta, 0, g));
    return gpx_sum_partials(h->stream, h->split.as<double>(), nsplit, stride, stride, C);
}

static int posterior_impl(gpx_t *h, const double *Xs, int64_t m, double *mu, double *s2,
                          double *dmu, double *ds2)
{
    CHECK_H(h);
    if (!h->have_factor) {
        gpx_set_error("gpx_exact_posterior: no factorisation (call gpx_exact_update)");
        return -1;
    }
    if (!Xs || !mu || !s2 || m < 0) {
        gpx_set_error("gpx_exact_posterior: bad arguments");
        return -1;
    }
    // test points per pass: as many as keep the two np x CH panels within ~2 GB
    // (one pass = one host synchronisation)
    const int CH = h->np <= 8192 ? 8192 : (h->np <= 16384 ? 4096 : 2048);
    const DenseWs w = h->ws();
    const bool grads = dmu && ds2;
    // V = R^-T K* is ONE triangle-aware product with W^T once W = R^-1 is complete;
    // with only the left-half inverses of a value-only update it is the recursive
    // solve of chol.hip (a chain of small launches, 1.5 ms at N = 16384 even for one
    // test point). Completing W costs a fraction of the factorisation, so do it when
    // there are many test points or when posterior calls repeat for this
    // factorisation (the acquisition loop of Bayesian optimisation), and always for
    // input gradients (alpha = W a and beta = W V need it).
    ++h->posterior_calls;
    if (!h->w_complete && (grads || h->posterior_calls >= 2 || m >= h->np / 4)) {
        GPX_TRY(gpx_trtri(h->stream, w));
        h->w_complete = true;
    }
    const bool by_gemm = h->w_complete;
    if (grads) {
        GPX_TRY(h->alpha.reserve((size_t)h->np * 8));
        GPX_TRY(gpx_trmv_upper(h->stream, w.W, h->ld, h->np, h->a.as<double>(),
                               h->alpha.as<double>()));
    }
    const double prior = gpx_kernel_prior(h->kp);
    StageClock clk(h);
    for (int64_t c0 = 0; c0 < m; c0 += CH) {
        const int mc = (int)std::min<int64_t>(CH, m - c0);
        const int mcp = round_up(mc, GPX_TILE);
        GPX_TRY(h->Xs.reserve((size_t)mc * h->d * 8));
        GPX_TRY(h->Ks.reserve((size_t)h->np * mcp * 8));
        GPX_TRY(h->KsT.reserve((size_t)h->np * mcp * 8));
        GPX_TRY(h->mu.reserve((size_t)mcp * 8));
        GPX_TRY(h->s2.reserve((size_t)mcp * 8));
        GPX_TRY(h->post_part.reserve(gpx_posterior_scratch(mcp) * 8));
        GPX_HIP(hipMemcpyAsync(h->Xs.p, Xs + c0 * h->d, (size_t)mc * h->d * 8,
                               hipMemcpyHostToDevice, h->stream));
        // K(X, Xs): np x mcp, zero outside n x mc (exact.py:87)
        GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                                   h->Xs.as<double>(), mc, mcp, h->d, h->Ks.as<double>(),
                                   mcp, false, false, 0.0));
        clk.tick(T_POST_BUILD);
        // RK = R^-T K (exact.py:88)
        double *V = h->Ks.as<double>();
        if (by_gemm) {
            GPX_TRY(tri_product(h, 1, h->Ks.as<double>(), h->KsT.as<double>(), mcp));
            V = h->KsT.as<double>();
        } else {
            GPX_TRY(gpx_trsm_rt(h->stream, w, h->Ks.as<double>(), h->KsT.as<double>(), mcp,
                                mcp));
        }
        GPX_TRY(gpx_posterior_reduce(h->stream, V, mcp, h->np, mcp,
                                     h->a.as<double>(), h->mean, prior,
                                     h->post_part.as<double>(), h->mu.as<double>(),
                                     h->s2.as<double>()));
        clk.tick(T_POST_SOLVE);
        if (grads) {
            // beta = W V (V = R^-T K*): W upper -> k >= row tile
            double *beta = (V == h->Ks.as<double>()) ? h->KsT.as<double>()
                                                     : h->Ks.as<double>();
            GPX_TRY(tri_product(h, 0, V, beta, mcp));
            GPX_TRY(h->t0.reserve((size_t)mc * h->d * 8));
            GPX_TRY(h->t1.reserve((size_t)mc * h->d * 8));
            GPX_TRY(h->gpart.reserve(gpx_posterior_grad_scratch(h->n, mc, h->d) * 8));
            GPX_TRY(gpx_posterior_grad(h->stream, h->kp, h->X.as<double>(), h->n,
                                       h->Xs.as<double>(), mc, h->d, h->alpha.as<double>(),
                                       beta, mcp, h->gpart.as<double>(), h->t0.as<double>(),
                                       h->t1.as<double>()));
            GPX_HIP(hipMemcpyAsync(dmu + c0 * h->d, h->t0.p, (size_t)mc * h->d * 8,
                                   hipMemcpyDeviceToHost, h->stream));
            GPX_HIP(hipMemcpyAsync(ds2 + c0 * h->d, h->t1.p, (size_t)mc * h->d * 8,
                                   hipMemcpyDeviceToHost, h->stream));
        }
        GPX_HIP(hipMemcpyAsync(mu + c0, h->mu.p, (size_t)mc * 8, hipMemcpyDeviceToHost,
                               h->stream));
        GPX_HIP(hipMemcpyAsync(s2 + c0, h->s2.p, (size_t)mc * 8, hipMemcpyDeviceToHost,
                               h->stream));
        GPX_HIP(hipStreamSynchronize(h->stream));
        if (c0 == 0) clk.collect();
    }
    return 0;
}

// ExactGP._full_posterior (exact.py:64-79): mean vector and FULL covariance
// Sigma = K(Xs, Xs) - V^T V, V = R^-T K(X, Xs); GP.sample draws from it
// (_base.py:143-178). One pass, m <= 8192.
int gpx_exact_posterior_full(gpx_t *h, const double *Xs, int64_t m, double *mu, double *Sigma)
{
    CHECK_H(h);
    if (!h->have_factor) {
        gpx_set_error("gpx_exact_posterior_full: no factorisation (call gpx_exact_update)");
        return -1;
    }
    if (!Xs || !mu || !Sigma || m < 1 || m > 8192) {
        gpx_set_error("gpx_exact_posterior_full: bad arguments (1 <= m <= 8192)");
        return -1;
    }
    const DenseWs w = h->ws();
    if (!h->w_complete) {
        GPX_TRY(gpx_trtri(h->stream, w));
        h->w_complete = true;
    }
    const int mc = (int)m, mcp = round_up(mc, GPX_TILE);
    GPX_TRY(h->Xs.reserve((size_t)mc * h->d * 8));
    GPX_TRY(h->Ks.reserve((size_t)h->np * mcp * 8));
    GPX_TRY(h->KsT.reserve((size_t)h->np * mcp * 8));
    GPX_TRY(h->mu.reserve((size_t)mcp * 8));
    GPX_TRY(h->s2.reserve((size_t)mcp * 8));
    GPX_TRY(h->post_part.reserve(gpx_posterior_scratch(mcp) * 8));
    GPX_TRY(h->t2.reserve((size_t)mcp * mcp * 8));
    GPX_HIP(hipMemcpyAsync(h->Xs.p, Xs, (size_t)mc * h->d * 8, hipMemcpyHostToDevice,
                           h->stream));
    GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                               h->Xs.as<double>(), mc, mcp, h->d, h->Ks.as<double>(), mcp,
                               false, false, 0.0));
    GPX_TRY(tri_product(h, 1, h->Ks.as<double>(), h->KsT.as<double>(), mcp));
    double *V = h->KsT.as<double>();
    GPX_TRY(gpx_posterior_reduce(h->stream, V, mcp, h->np, mcp, h->a.as<double>(), h->mean,
                                 0.0, h->post_part.as<double>(), h->mu.as<double>(),
                                 h->s2.as<double>()));
    // Sigma = K(Xs, Xs) - V^T V on the padded mcp x mcp block
    GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->Xs.as<double>(), mc, mcp,
                               h->Xs.as<double>(), mc, mcp, h->d, h->t2.as<double>(), mcp,
                               false, false, 0.0));
    GemmArgs g;
    g.A = V; g.B = V; g.C = h->t2.as<double>();
    g.lda = mcp; g.ldb = mcp; g.ldc = mcp;
    g.M = mcp; g.N = mcp; g.K = h->np;
    g.alpha = -1.0; g.beta = 1.0;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = 0;
    g.tile = 0; g.order = 0; g.swizzle = 0; g.waves = 0; g.use_lists = 1;
    g.tiles = nullptr;
    GPX_TRY(gpx_gemm(h->stream, 1, 0, g));
    GPX_HIP(hipMemcpyAsync(mu, h->mu.p, (size_t)mc * 8, hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipMemcpy2DAsync(Sigma, (size_t)mc * 8, h->t2.p, (size_t)mcp * 8, (size_t)mc * 8, mc,
                             hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int gpx_exact_posterior(gpx_t *h, const double *Xs, int64_t m, double *mu, double *s2)
{
    return posterior_impl(h, Xs, m, mu, s2, nullptr, nullptr);
}

int gpx_exact_posterior_grad(gpx_t *h, const double *Xs, int64_t m, double *mu, double *s2,
                             double *dmu, double *ds2)
{
    if (!dmu || !ds2) {
        gpx_set_error("gpx_exact_posterior_grad: null gradient outputs");
        return -1;
    }
    return posterior_impl(h, Xs, m, mu, s2, dmu, ds2);
}

// [m.posterior(X, grad) for m in samples] of the reference's meta-models
// (meta/mcmc.py:75-77, meta/smc.py:128-130): B models that share the resident
// data and differ in their hyperparameters. The update of model b+1.. is already
// running on another context while the host waits for the posterior of model b.
int gpx_posterior_batch(gpx_t *h, const gpx_kspec *k, const double *thetas, int64_t B,
                        const double *Xs, int64_t m, double *mu, double *s2, double *dmu,
                        double *ds2, int *info)
{
    CHECK_H(h);
    if (!k || !thetas || !Xs || !mu || !s2 || B < 0 || m < 0 || (!dmu) != (!ds2)) {
        gpx_set_error("gpx_posterior_batch: bad arguments");
        return -1;
    }
    if (h->n <= 0) {
        gpx_set_error("no data: call gpx_set_data first");
        return -1;
    }
    const int nth = 1 + k->nhyper + 1;
    const bool grads = dmu && ds2;
    // groups of members in lock-step, as gpx_loglik_batch (group.hip; round 4)
    if (m > 0 && batch_in_groups(h, B)) {
        const int rc = gpx_groups_posterior(&h->groups, h->device, h->X.as<double>(),
                                            h->y.as<double>(), h->n, h->d, h->np, k, thetas, B,
                                            Xs, m, grads, mu, s2, dmu, ds2, info);
        if (rc != 1) {
            h->have_factor = h->have_inverse = false;
            return rc;
        }
    }
    int depth = (int)std::min<int64_t>(3, std::max<int64_t>(B, 1));
    const double ws_bytes = 3.0 * h->np * (double)h->ld * 8;
    while (depth > 1 && depth * ws_bytes > 160e9) --depth;
    gpx_ctx *ctx[3] = {h, h, h};
    for (int i = 1; i < depth; ++i) {
        GPX_TRY(ensure_twin(ctx[i - 1]));
        ctx[i] = ctx[i - 1]->twin;
    }
    const bool timing = h->timing;
    h->timing = false;
    std::vector<gpx_kspec> store[3];
    int rc = 0;
    if (depth > 1) gpx_gemm_concurrency(h->device, +1);
    for (int i = 0; i < depth; ++i) ctx[i]->in_batch = depth > 1;   // (one stream each: ws())
    auto start = [&](int64_t b) -> int {          // hypers + K + Cholesky + a, no sync
        gpx_ctx *c = ctx[b % depth];
        const double *th = thetas + b * nth;
        gpx_kspec kb;
        GPX_TRY(gpx_kspec_with_hyper(k, th + 1, store[b % depth], &kb));
        GPX_HIP(hipSetDevice(c->device));
        GPX_TRY(check_ready(c, &kb, th[0], th[nth - 1]));
        GPX_TRY(reserve_factor(c, grads));
        c->have_factor = c->have_inverse = false;
        StageClock clk(c);
        GPX_TRY(enqueue_update(c, clk, grads ? GPX_POTRF_W : GPX_POTRF_R));
        GPX_HIP(hipMemcpyAsync(c->hinfo, c->info.p, sizeof(int), hipMemcpyDeviceToHost,
                               c->stream));
        c->have_factor = true;                    // checked through hinfo in finish()
        return 0;
    };
    auto finish_one = [&](int64_t b) -> int {
        gpx_ctx *c = ctx[b % depth];
        double *mub = mu + b * m, *s2b = s2 + b * m;
        double *dmub = grads ? dmu + b * m * c->d : nullptr;
        double *ds2b = grads ? ds2 + b * m * c->d : nullptr;
        int r = 0;
        if (m > 0) r = posterior_impl(c, Xs, m, mub, s2b, dmub, ds2b);
        else GPX_HIP(hipStreamSynchronize(c->stream));
        if (r < 0) return r;
        const int inf = *c->hinfo;
        if (info) info[b] = inf;
        if (inf < 0) {
            gpx_set_error("internal: the panel kernel timed out waiting for a dependency");
            return -1;
        }
        if (inf > 0) {                            // not PD: this model has no posterior
            c->have_factor = false;
            for (int64_t i = 0; i < m; ++i) mub[i] = s2b[i] = NAN;
            if (grads)
                for (int64_t i = 0; i < m * c->d; ++i) dmub[i] = ds2b[i] = NAN;
        }
        return 0;
    };
    for (int64_t b = 0; b < B && rc >= 0; ++b) {
        rc = start(b);
        if (rc >= 0 && b >= depth - 1) rc = finish_one(b - (depth - 1));
    }
    for (int64_t b = std::max<int64_t>(0, B - (depth - 1)); b < B && rc >= 0; ++b)
        rc = finish_one(b);
    if (depth > 1) gpx_gemm_concurrency(h->device, -1);
    for (int i = 0; i < depth; ++i) ctx[i]->in_batch = false;
    (void)hipSetDevice(h->device);
    h->timing = timing;
    h->have_factor = h->have_inverse = false;          // as in gpx_loglik_batch
    return rc < 0 ? rc : 0;
}

int gpx_kernel_gradx(gpx_t *h, const gpx_kspec *k, const double *X1, int64_t n1,
                     const double *X2, int64_t n2, int64_t d, int wrt, double *out)
{
    CHECK_H(h);
    if (!X1 || !out || n1 < 0 || (X2 && n2 < 0) || (wrt != 1 && wrt != 2)) {
        gpx_set_error("gpx_kernel_gradx: bad arguments");
        return -1;
    }
    if (!X2) n2 = n1;
    if (n1 == 0 || n2 == 0) return 0;
    KParams kp;
    GPX_TRY(gpx_flatten_kspec(k, d, &kp));
    const size_t xb1 = (size_t)n1 * d * 8, xb2 = (size_t)n2 * d * 8;
    const size_t ob = (size_t)n1 * n2 * d * 8;
    GPX_TRY(h->t0.reserve(xb1));
    GPX_TRY(h->t2.reserve(ob));
    GPX_HIP(hipMemcpyAsync(h->t0.p, X1, xb1, hipMemcpyHostToDevice, h->stream));
    const double *dX2 = h->t0.as<double>();
    if (X2) {
        GPX_TRY(h->t1.reserve(xb2));
        GPX_HIP(hipMemcpyAsync(h->t1.p, X2, xb2, hipMemcpyHostToDevice, h->stream));
        dX2 = h->t1.as<double>();
    }
    GPX_TRY(gpx_kgrady(h->stream, kp, h->t0.as<double>(), (int)n1, dX2, (int)n2, (int)d,
                       wrt == 2 ? 1.0 : -1.0, h->t2.as<double>()));
    GPX_HIP(hipMemcpyAsync(out, h->t2.p, ob, hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int gpx_exact_get_factor(gpx_t *h, int64_t n, double *R, double *a)
{
    CHECK_H(h);
    if (!h->have_factor) {
        gpx_set_error("gpx_exact_get_factor: no factorisation");
        return -1;
    }
    if (n != h->n) {
        gpx_set_error("gpx_exact_get_factor: caller expects %lld points, the factor has %d",
                      (long long)n, h->n);
        return -1;
    }
    if (R) {
        const size_t bytes = (size_t)h->n * h->n * 8;
        GPX_TRY(h->t2.reserve(bytes));
        GPX_TRY(gpx_copy_upper(h->stream, h->A.as<double>(), h->ld, h->n,
                               h->t2.as<double>()));
        GPX_HIP(hipMemcpyAsync(R, h->t2.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    if (a)
        GPX_HIP(hipMemcpyAsync(a, h->a.p, (size_t)h->n * 8, hipMemcpyDeviceToHost,
                               h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- dense building blocks ---------------------------------------------------
int gpx_la_gemm(gpx_t *h, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                const double *A, int64_t lda, const double *B, int64_t ldb, double beta,
                double *C, int64_t ldc)
{
    CHECK_H(h);
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) {
        gpx_set_error("gpx_la_gemm: bad arguments");
        return -1;
    }
    const int Mp = round_up(M, GPX_TILE), Np = round_up(N, GPX_TILE),
              Kp = round_up(K, GPX_TILE);
    // stored shapes: A is (ta ? K x M : M x K), B is (tb ? N x K : K x N)
    const int ar = ta ? Kp : Mp, ac = ta ? Mp : Kp, br = tb ? Np : Kp, bc = tb ? Kp : Np;
    const int64_t har = ta ? K : M, hac = ta ? M : K, hbr = tb ? N : K, hbc = tb ? K : N;
    GPX_TRY(h->t0.reserve((size_t)ar * ac * 8));
    GPX_TRY(h->t1.reserve((size_t)br * bc * 8));
    GPX_TRY(h->t2.reserve((size_t)Mp * Np * 8));
    GPX_HIP(hipMemsetAsync(h->t0.p, 0, (size_t)ar * ac * 8, h->stream));
    GPX_HIP(hipMemsetAsync(h->t1.p, 0, (size_t)br * bc * 8, h->stream));
    GPX_HIP(hipMemsetAsync(h->t2.p, 0, (size_t)Mp * Np * 8, h->stream));
    GPX_HIP(hipMemcpy2DAsync(h->t0.p, (size_t)ac * 8, A, (size_t)lda * 8, (size_t)hac * 8,
                             har, hipMemcpyHostToDevice, h->stream));
    GPX_HIP(hipMemcpy2DAsync(h->t1.p, (size_t)bc * 8, B, (size_t)ldb * 8, (size_t)hbc * 8,
                             hbr, hipMemcpyHostToDevice, h->stream));
    if (beta != 0.0)
        GPX_HIP(hipMemcpy2DAsync(h->t2.p, (size_t)Np * 8, C, (size_t)ldc * 8,
                                 (size_t)N * 8, M, hipMemcpyHostToDevice, h->stream));
    GemmArgs g;
    g.A = h->t0.as<double>(); g.B = h->t1.as<double>(); g.C = h->t2.as<double>();
    g.lda = ac; g.ldb = bc; g.ldc = Np;
    g.M = Mp; g.N = Np; g.K = Kp;
    g.alpha = alpha; g.beta = beta;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = 0;
    g.tile = 0;
    g.order = 0;
    g.swizzle = 0;
    g.waves = 0;
    g.use_lists = 1;
    g.tiles = nullptr;
    GPX_TRY(gpx_gemm(h->stream, ta, tb, g));
    GPX_HIP(hipMemcpy2DAsync(C, (size_t)ldc * 8, h->t2.p, (size_t)Np * 8, (size_t)N * 8, M,
                             hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

__global__ void pad_identity_kernel(double *__restrict__ A, int ld, int np, int n)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < np && (i >= n || j >= n)) A[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

int gpx_la_potrf(gpx_t *h, const double *A, int64_t n, double *R, double *Rinv,
                 double *Ainv, int *info)
{
    CHECK_H(h);
    if (!A || n < 1 || n > (1 << 20)) {
        gpx_set_error("gpx_la_potrf: bad arguments");
        return -1;
    }
    h->have_factor = h->have_inverse = false;
    h->n = 0;                                   // the GP state is gone
    h->np = h->cap = round_up(n, GPX_TILE);
    h->ld = ld_for(h->np);
    const int np = h->np, ld = h->ld;
    GPX_TRY(reserve_factor(h, true));
    const DenseWs w = h->ws();
    GPX_HIP(hipMemsetAsync(h->info.p, 0, sizeof(int), h->stream));
    GPX_HIP(hipMemcpy2DAsync(w.A, (size_t)ld * 8, A, (size_t)n * 8, (size_t)n * 8, n,
                             hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(pad_identity_kernel, dim3((np + 255) / 256, np), dim3(256), 0,
                       h->stream, w.A, ld, np, (int)n);
    const int la_mode = Ainv ? GPX_POTRF_KINV : (Rinv ? GPX_POTRF_W : GPX_POTRF_R);
    DenseWs wl = w;
    wl.whole = gpx_potrf_whole(wl, la_mode);
    GPX_TRY(gpx_potrf(h->stream, wl, la_mode, false));
    const size_t bytes = (size_t)n * n * 8;
    GPX_TRY(h->t2.reserve(bytes));
    if (R) {
        GPX_TRY(gpx_copy_upper(h->stream, w.A, ld, (int)n, h->t2.as<double>()));
        GPX_HIP(hipMemcpyAsync(R, h->t2.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    if (Rinv) {
        GPX_TRY(gpx_copy_upper(h->stream, w.W, ld, (int)n, h->t2.as<double>()));
        GPX_HIP(hipMemcpyAsync(Rinv, h->t2.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    if (Ainv) {
        GPX_TRY(gpx_symmetrize(h->stream, w.Kinv, ld, (int)n, h->t2.as<double>()));
        GPX_HIP(hipMemcpyAsync(Ainv, h->t2.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    int inf = 0;
    GPX_HIP(hipMemcpyAsync(&inf, h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    if (info) *info = inf;
    if (inf) {
        gpx_set_error("matrix is not positive definite: pivot %d", inf);
        return inf;
    }
    return 0;
}

__global__ void fill_uniform_kernel(double *__restrict__ p, size_t n, unsigned seed)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int gpx_la_gemm_bench_ex(gpx_t *h, int ta, int tb, int64_t n, int flags, int order,
                         int swizzle, int tile, int waves, int same_ab, int reps,
                         double *ms)
{
    CHECK_H(h);
    if (n < 1 || n % GPX_TILE) {
        gpx_set_error("gpx_la_gemm_bench: n must be a multiple of %d", GPX_TILE);
        return -1;
    }
    // GPX_BENCH_LDPAD: row stride n + pad, as the factorisation workspaces have
    static const int ldpad = [] {
        const char *e = getenv("GPX_BENCH_LDPAD");
        const int v = e ? atoi(e) : 0;
        return v < 0 || v % 2 ? 0 : v;
    }();
    const size_t ldn = (size_t)n + ldpad;
    const size_t cnt = (size_t)n * ldn;
    const bool fresh = h->t0.bytes < cnt * 8 || h->t1.bytes < cnt * 8 ||
                       h->bench_n != n;
    GPX_TRY(h->t0.reserve(cnt * 8));
    GPX_TRY(h->t1.reserve(cnt * 8));
    GPX_TRY(h->t2.reserve(cnt * 8));
    if (fresh) {
        const unsigned blocks = (unsigned)((cnt + 255) / 256);
        hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, h->stream,
                           h->t0.as<double>(), cnt, 1u);
        hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, h->stream,
                           h->t1.as<double>(), cnt, 2u);
        h->bench_n = n;
    }
    GemmArgs g;
    g.A = h->t0.as<double>(); g.B = h->t1.as<double>(); g.C = h->t2.as<double>();
    if (same_ab) g.B = g.A;
    g.lda = g.ldb = g.ldc = (int)ldn;
    g.M = g.N = g.K = (int)n;
    g.alpha = 1.0; g.beta = 0.0;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = flags;
    g.tile = tile;
    g.order = order;
    g.swizzle = swizzle & 1;
    g.waves = waves;
    g.use_lists = (swizzle & 2) ? 0 : 1;      // bit 1 of swizzle: plain 2-D grid
    g.tiles = nullptr;
    if (reps < 1) reps = 1;
    hipEvent_t e0 = h->ev[GPX_NTIMERS], e1 = h->ev[0];
    GPX_HIP(hipEventRecord(e0, h->stream));
    for (int it = 0; it < reps; ++it) GPX_TRY(gpx_gemm(h->stream, ta, tb, g));
    GPX_HIP(hipEventRecord(e1, h->stream));
    GPX_HIP(hipEventSynchronize(e1));
    float t = 0;
    GPX_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / reps;
    return 0;
}

// device-resident timing of one M x N x K product with the engine's structure flags:
// C (M x N) = alpha op(A) op(B) + beta C on synthetic operands (beta != 0: the rank-K
// update shape of the factorisation)
int gpx_la_gemm_bench_mnk(gpx_t *h, int ta, int tb, int64_t M, int64_t N, int64_t K, int flags,
                          double beta, int tile, int reps, double *ms)
{
    CHECK_H(h);
    if (M < 1 || N < 1 || K < 1 || M % GPX_TILE || N % GPX_TILE || K % GPX_TILE) {
        gpx_set_error("gpx_la_gemm_bench_mnk: M, N, K must be multiples of %d", GPX_TILE);
        return -1;
    }
    const size_t big = (size_t)std::max<int64_t>(M, std::max<int64_t>(N, K));
    const size_t ldn = big + 32;
    const size_t cnt = big * ldn;
    GPX_TRY(h->t0.reserve(cnt * 8));
    GPX_TRY(h->t1.reserve(cnt * 8));
    GPX_TRY(h->t2.reserve(cnt * 8));
    const unsigned blocks = (unsigned)((cnt + 255) / 256);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, h->stream,
                       h->t0.as<double>(), cnt, 1u);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(blocks), dim3(256), 0, h->stream,
                       h->t1.as<double>(), cnt, 2u);
    GPX_HIP(hipMemsetAsync(h->t2.p, 0, cnt * 8, h->stream));
    h->bench_n = 0;
    GemmArgs g;
    g.A = h->t0.as<double>(); g.B = h->t1.as<double>(); g.C = h->t2.as<double>();
    g.lda = g.ldb = g.ldc = (int)ldn;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.alpha = beta != 0.0 ? -1e-6 : 1.0; g.beta = beta;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = flags;
    g.tile = tile;
    g.order = 0;
    g.swizzle = 0;
    g.waves = 0;
    g.use_lists = 1;
    g.tiles = nullptr;
    if (reps < 1) reps = 1;
    hipEvent_t e0 = h->ev[GPX_NTIMERS], e1 = h->ev[0];
    GPX_TRY(gpx_gemm(h->stream, ta, tb, g));                 // warm-up (tile lists)
    GPX_HIP(hipEventRecord(e0, h->stream));
    for (int it = 0; it < reps; ++it) GPX_TRY(gpx_gemm(h->stream, ta, tb, g));
    GPX_HIP(hipEventRecord(e1, h->stream));
    GPX_HIP(hipEventSynchronize(e1));
    float t = 0;
    GPX_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / reps;
    return 0;
}

int gpx_la_gemm_bench(gpx_t *h, int ta, int tb, int64_t n, int reps, double *ms)
{
    double warm = 0;
    GPX_TRY(gpx_la_gemm_bench_ex(h, ta, tb, n, 0, 0, 0, 0, 0, 0, 1, &warm));
    return gpx_la_gemm_bench_ex(h, ta, tb, n, 0, 0, 0, 0, 0, 0, reps, ms);
}

int gpx_la_potrf_bench(gpx_t *h, int64_t n, int with_inverse, int reps, double *ms)
{
    CHECK_H(h);
    if (n < 1 || n > (1 << 20)) {
        gpx_set_error("gpx_la_potrf_bench: bad n");
        return -1;
    }
    // synthetic SPD matrix: SE kernel on uniform points + 0.01 I
    const int d = 8;
    h->have_factor = h->have_inverse = false;
    h->n = (int)n;
    h->d = d;
    h->np = h->cap = round_up(n, GPX_TILE);
    h->ld = ld_for(h->np);
    GPX_TRY(h->X.reserve((size_t)n * d * 8));
    GPX_TRY(h->y.reserve((size_t)n * 8));
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256),
                       0, h->stream, h->X.as<double>(), (size_t)n * d, 7u);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       h->stream, h->y.as<double>(), (size_t)n, 8u);
    double hyper[1 + d];
    hyper[0] = 0.0;
    for (int i = 0; i < d; ++i) hyper[1 + i] = 0.0;
    gpx_kspec k = {GPX_SE, 0, d, 1 + d, hyper, 0, nullptr};
    GPX_TRY(gpx_flatten_kspec(&k, d, &h->kp));
    GPX_TRY(reserve_factor(h, with_inverse != 0));
    const DenseWs w = h->ws();
    if (reps < 1) reps = 1;
    double total = 0.0;
    hipEvent_t e0 = h->ev[GPX_NTIMERS], e1 = h->ev[0];
    for (int it = -1; it < reps; ++it) {
        GPX_HIP(hipMemsetAsync(h->info.p, 0, sizeof(int), h->stream));
        GPX_TRY(gpx_kbuild<double>(h->stream, h->kp, h->X.as<double>(), h->n, h->np,
                                   h->X.as<double>(), h->n, h->np, d, w.A, h->ld, true,
                                   true, 0.01, w.Kinv));
        GPX_HIP(hipEventRecord(e0, h->stream));
        DenseWs wl = w;
        wl.whole = gpx_potrf_whole(wl, with_inverse ? GPX_POTRF_KINV : GPX_POTRF_R);
        GPX_TRY(gpx_potrf(h->stream, wl, with_inverse ? GPX_POTRF_KINV : GPX_POTRF_R, true));
        GPX_HIP(hipEventRecord(e1, h->stream));
        GPX_HIP(hipEventSynchronize(e1));
        float t = 0;
        GPX_HIP(hipEventElapsedTime(&t, e0, e1));
        if (it >= 0) total += t;
    }
    int inf = 0;
    GPX_HIP(hipMemcpy(&inf, h->info.p, sizeof(int), hipMemcpyDeviceToHost));
    if (inf) {
        gpx_set_error("gpx_la_potrf_bench: synthetic matrix not PD at %d", inf);
        return inf;
    }
    if (ms) *ms = total / reps;
    return 0;
}

}  // extern "C"
