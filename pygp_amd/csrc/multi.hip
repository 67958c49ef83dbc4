// Batched hyperparameter evaluation (log-likelihoods + gradients, and posteriors) over
// the GPUs of one node from ONE process: the torch-free multi-device entries of the
// C ABI (include/gpx.h).
//
// The reference's only data-parallel structure is its Python loops over independent
// hyperparameter samples (pygp/meta/smc.py:102-126, pygp/meta/mcmc.py:75-77,
// pygp/learning/sampling.py:146 under /root/reference). Here B (theta, shared
// dataset) evaluations are block-partitioned over ndev devices; every device has
// its own handle (stream + workspace, X and y replicated) driven by its own host
// thread through gpx_loglik_batch, and the per-member results are assembled with
// ONE ncclAllGather (RCCL over xGMI) on a single-process ncclCommInitAll
// communicator. A single GP never spans GPUs; ndev = 1 needs no RCCL at all, and
// librccl.so is only loaded (dlopen) the first time ndev > 1 is asked for.

#include "gpx_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct Pool {
    std::mutex mu;                     // one multi-device call at a time
    std::vector<gpx_t *> handle;       // one per device, created on first use
    Rccl rccl;
    int comm_ndev = 0;                 // devices of the cached communicator
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;   // collective streams, one per device
    std::vector<double *> send, recv;  // device staging of the gather
    std::vector<size_t> send_cap, recv_cap;   // doubles, per device (0: not allocated)
    int64_t res_n = 0, res_d = 0;      // data resident on the first res_ndev handles
    int res_ndev = 0;
};

Pool &pool()
{
    static Pool p;
    return p;
}

int load_rccl(Rccl &r)
{
    if (r.lib) return 0;
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        gpx_set_error("multi-device gather: cannot load librccl.so (%s)", dlerror());
        return -2;
    }
#define GPX_SYM(field, name)                                                       \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));             \
    if (!r.field) {                                                                \
        gpx_set_error("multi-device gather: librccl.so lacks %s", name);           \
        return -2;                                                                 \
    }
    GPX_SYM(CommInitAll, "ncclCommInitAll")
    GPX_SYM(CommDestroy, "ncclCommDestroy")
    GPX_SYM(AllGather, "ncclAllGather")
    GPX_SYM(GroupStart, "ncclGroupStart")
    GPX_SYM(GroupEnd, "ncclGroupEnd")
    GPX_SYM(GetErrorString, "ncclGetErrorString")
#undef GPX_SYM
    return 0;
}

#define GPX_NCCL(p, expr)                                                          \
    do {                                                                           \
        ncclResult_t r_ = (expr);                                                  \
        if (r_ != ncclSuccess) {                                                   \
            gpx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                          (p).rccl.GetErrorString(r_));                            \
            return -2;                                                             \
        }                                                                          \
    } while (0)

int ensure_comm(Pool &p, int ndev)
{
    GPX_TRY(load_rccl(p.rccl));
    if (p.comm_ndev == ndev) return 0;
    for (ncclComm_t c : p.comm) (void)p.rccl.CommDestroy(c);
    p.comm.assign(ndev, nullptr);
    p.comm_ndev = 0;
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) devs[i] = i;
    GPX_NCCL(p, p.rccl.CommInitAll(p.comm.data(), ndev, devs.data()));
    p.comm_ndev = ndev;
    for (int i = (int)p.stream.size(); i < ndev; ++i) {
        hipStream_t s = nullptr;
        GPX_HIP(hipSetDevice(i));
        GPX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        p.stream.push_back(s);
        p.send.push_back(nullptr);
        p.recv.push_back(nullptr);
        p.send_cap.push_back(0);
        p.recv_cap.push_back(0);
    }
    return 0;
}

// staging of the first ndev devices, each tracked on its own: a call with more devices
// than the one before (a strong-scaling sweep in one process) must allocate for the
// new ones even when the per-device sizes shrink
int ensure_staging(Pool &p, int ndev, size_t send_doubles)
{
    const size_t recv_doubles = send_doubles * ndev;
    for (int i = 0; i < ndev; ++i) {
        if (p.send[i] && p.recv[i] && send_doubles <= p.send_cap[i] &&
            recv_doubles <= p.recv_cap[i])
            continue;
        GPX_HIP(hipSetDevice(i));
        if (p.send[i]) GPX_HIP(hipFree(p.send[i]));
        if (p.recv[i]) GPX_HIP(hipFree(p.recv[i]));
        p.send[i] = p.recv[i] = nullptr;
        p.send_cap[i] = p.recv_cap[i] = 0;
        GPX_HIP(hipMalloc((void **)&p.send[i], send_doubles * 8));
        GPX_HIP(hipMalloc((void **)&p.recv[i], recv_doubles * 8));
        p.send_cap[i] = send_doubles;
        p.recv_cap[i] = recv_doubles;
    }
    return 0;
}

}  // namespace

extern "C" {

void gpx_batch_partition(int64_t B, int world, int rank, int64_t *lo, int64_t *hi)
{
    // contiguous blocks; the first B % world ranks take one extra member
    // (pygp_amd/batch.py partition() is the same rule)
    const int64_t base = B / world, extra = B % world;
    const int64_t a = rank * base + (rank < extra ? rank : extra);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < extra ? 1 : 0);
}

// Host halves of the gather (no device, no RCCL: unit-tested on CPU). Every device
// sends one slot of `slot = ceil(B / ndev)` member rows of `width` doubles; a device
// with fewer members (or none, B < ndev) pads with NaN, which the scatter never reads.
int64_t gpx_multi_slot(int64_t B, int ndev) { return ndev > 0 ? (B + ndev - 1) / ndev : 0; }

int gpx_multi_comm_size(void)
{
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    return p.comm_ndev;
}

int gpx_multi_pack(const double *rows, int64_t cnt, int width, int64_t slot, double *pack)
{
    if (cnt < 0 || cnt > slot || width < 1 || !pack || (cnt > 0 && !rows)) {
        gpx_set_error("gpx_multi_pack: bad arguments");
        return -1;
    }
    const size_t used = (size_t)cnt * width, all = (size_t)slot * width;
    if (used) memcpy(pack, rows, used * 8);
    for (size_t i = used; i < all; ++i) pack[i] = NAN;
    return 0;
}

int gpx_multi_scatter(const double *gathered, int64_t B, int ndev, int width, double *out)
{
    if (B < 0 || ndev < 1 || width < 1 || (B > 0 && (!gathered || !out))) {
        gpx_set_error("gpx_multi_scatter: bad arguments");
        return -1;
    }
    const int64_t slot = gpx_multi_slot(B, ndev);
    for (int dev = 0; dev < ndev; ++dev) {
        int64_t lo, hi;
        gpx_batch_partition(B, ndev, dev, &lo, &hi);
        if (hi > lo)
            memcpy(out + (size_t)lo * width, gathered + (size_t)dev * slot * width,
                   (size_t)(hi - lo) * width * 8);
    }
    return 0;
}

}  // extern "C"

// B members dealt to ndev devices, each member producing `width` doubles. run(dev, h, lo,
// cnt, rows): evaluate members [lo, lo + cnt) on device dev's handle h into rows
// [cnt][width]; returns the C-ABI code of the call it makes. The rows of all members come
// back in out[B][width]: directly for one device, through ONE ncclAllGather otherwise.
template <typename Run>
static int multi_run(const char *what, int ndev, int64_t B, int width, const double *X,
                     const double *y, int64_t n, int64_t d, double *out, Run run)
{
    int have = 0;
    GPX_HIP(hipGetDeviceCount(&have));
    // GPX_MULTI_FAKE=1 (rehearsal on a box with fewer GPUs than ranks): logical device i is
    // physical device i % present -- its own handle, its own host thread, the same
    // partition, packing and scatter -- and the ONE collective is replaced by the
    // concatenation ncclAllGather would deliver (RCCL refuses a communicator with one GPU
    // twice). Everything but the wire is exercised; never set in production.
    static const bool fake = getenv("GPX_MULTI_FAKE") && atoi(getenv("GPX_MULTI_FAKE"));
    if (have < 1 || (ndev > have && !fake)) {
        gpx_set_error("%s: %d devices asked for, %d present", what, ndev, have);
        return -1;
    }
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (!X && (p.res_n < 1 || p.res_ndev < ndev)) {
        gpx_set_error("%s: X == NULL but no data is resident on %d devices (pass X, y once)",
                      what, ndev);
        return -1;
    }
    while ((int)p.handle.size() < ndev) p.handle.push_back(nullptr);
    for (int i = 0; i < ndev; ++i)
        if (!p.handle[i]) GPX_TRY(gpx_create(i % have, &p.handle[i]));

    // GPX_MULTI_FORCE_RCCL=1 sends a one-device call through the collective too (the
    // rehearsal of the gather on a one-GPU box)
    static const bool force = getenv("GPX_MULTI_FORCE_RCCL") && atoi(getenv("GPX_MULTI_FORCE_RCCL"));
    const bool gather = ndev > 1 || force;
    const int64_t slot = gpx_multi_slot(B, ndev);            // members per device slot
    std::vector<std::vector<double>> loc(ndev);
    std::vector<int> rc(ndev, 0);
    std::vector<std::string> err(ndev);

    auto work = [&](int dev) {
        int64_t lo, hi;
        gpx_batch_partition(B, ndev, dev, &lo, &hi);
        const int64_t cnt = hi - lo;
        loc[dev].assign((size_t)cnt * width, 0.0);
        gpx_t *h = p.handle[dev];
        int r = X ? gpx_set_data(h, X, n, d, y) : 0;          // NULL: keep the resident data
        if (r == 0 && cnt > 0) r = run(dev, h, lo, cnt, loc[dev].data());
        rc[dev] = r;
        if (r != 0) err[dev] = gpx_last_error();              // thread-local text
    };
    if (ndev == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int dev = 0; dev < ndev; ++dev) th.emplace_back(work, dev);
        for (std::thread &t : th) t.join();
    }
    if (X) {
        p.res_n = n;
        p.res_d = d;
        p.res_ndev = ndev;
    }
    for (int dev = 0; dev < ndev; ++dev)
        if (rc[dev] != 0) {
            if (X) p.res_n = 0;
            gpx_set_error("device %d: %s", dev, err[dev].c_str());
            return rc[dev] < 0 ? rc[dev] : -2;
        }

    if (gather && B > 0 && fake && ndev > have) {
        std::vector<double> all((size_t)ndev * slot * width);     // what the gather delivers
        for (int dev = 0; dev < ndev; ++dev)
            GPX_TRY(gpx_multi_pack(loc[dev].data(), (int64_t)(loc[dev].size() / width), width,
                                   slot, all.data() + (size_t)dev * slot * width));
        GPX_TRY(gpx_multi_scatter(all.data(), B, ndev, width, out));
    } else if (gather && B > 0) {
        std::vector<double> all;                              // [ndev][slot][width]
        GPX_TRY(ensure_comm(p, ndev));
        GPX_TRY(ensure_staging(p, ndev, (size_t)slot * width));
        std::vector<double> pack((size_t)slot * width);
        for (int dev = 0; dev < ndev; ++dev) {
            GPX_TRY(gpx_multi_pack(loc[dev].data(), (int64_t)(loc[dev].size() / width), width,
                                   slot, pack.data()));
            GPX_HIP(hipSetDevice(dev));
            GPX_HIP(hipMemcpyAsync(p.send[dev], pack.data(), pack.size() * 8,
                                   hipMemcpyHostToDevice, p.stream[dev]));
            GPX_HIP(hipStreamSynchronize(p.stream[dev]));     // pack is reused
        }
        // the one collective of the batched-theta path; the group is always closed, also
        // when a member call fails (an open group would swallow every later collective)
        GPX_NCCL(p, p.rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int dev = 0; dev < ndev && bad == ncclSuccess; ++dev)
            bad = p.rccl.AllGather(p.send[dev], p.recv[dev], (size_t)slot * width, ncclDouble,
                                   p.comm[dev], p.stream[dev]);
        const ncclResult_t end = p.rccl.GroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) {
            gpx_set_error("%s: ncclAllGather -> %s", what,
                          p.rccl.GetErrorString(bad != ncclSuccess ? bad : end));
            return -2;
        }
        for (int dev = 0; dev < ndev; ++dev) {
            GPX_HIP(hipSetDevice(dev));
            GPX_HIP(hipStreamSynchronize(p.stream[dev]));
        }
        all.resize((size_t)ndev * slot * width);
        GPX_HIP(hipSetDevice(0));
        GPX_HIP(hipMemcpy(all.data(), p.recv[0], all.size() * 8, hipMemcpyDeviceToHost));
        GPX_TRY(gpx_multi_scatter(all.data(), B, ndev, width, out));
    } else {
        for (int dev = 0; dev < ndev; ++dev) {
            int64_t lo, hi;
            gpx_batch_partition(B, ndev, dev, &lo, &hi);
            if (hi > lo) memcpy(out + (size_t)lo * width, loc[dev].data(), loc[dev].size() * 8);
        }
    }
    (void)hipSetDevice(0);
    return 0;
}

extern "C" {

int gpx_loglik_batch_multi(const gpx_kspec *k, const double *thetas, int64_t B,
                           const double *X, const double *y, int64_t n, int64_t d,
                           int want_grad, int ndev, double *lZ, double *dlZ, int *info)
{
    if (!k || !thetas || !lZ || B < 0 || ndev < 1 || (!X) != (!y) ||
        (X && (n < 1 || d < 1))) {
        gpx_set_error("gpx_loglik_batch_multi: bad arguments");
        return -1;
    }
    const int nth = 1 + k->nhyper + 1;
    const bool grad = want_grad && dlZ;
    // per member: [lZ | dlZ (nth, if grad) | info]
    const int width = 1 + (grad ? nth : 0) + 1;
    std::vector<double> rows((size_t)B * width);
    GPX_TRY(multi_run("gpx_loglik_batch_multi", ndev, B, width, X, y, n, d, rows.data(),
                      [&](int, gpx_t *h, int64_t lo, int64_t cnt, double *out) -> int {
                          std::vector<double> l(cnt), g(grad ? cnt * nth : 0);
                          std::vector<int> inf(cnt, 0);
                          const int r = gpx_loglik_batch(h, k, thetas + lo * nth, cnt, grad ? 1 : 0,
                                                         l.data(), grad ? g.data() : nullptr,
                                                         inf.data());
                          for (int64_t b = 0; b < cnt; ++b) {
                              double *row = out + b * width;
                              row[0] = l[b];
                              if (grad) memcpy(row + 1, g.data() + b * nth, nth * 8);
                              row[width - 1] = (double)inf[b];
                          }
                          return r;
                      }));
    for (int64_t b = 0; b < B; ++b) {
        const double *row = rows.data() + b * width;
        lZ[b] = row[0];
        if (grad) memcpy(dlZ + b * nth, row + 1, nth * 8);
        if (info) info[b] = (int)row[width - 1];
    }
    return 0;
}

// Audit entries (include/gpx.h): the per-device handles live inside this file, so a bench
// that drives N devices from one process reads the event time of every device's groups, how
// that device cut its block and whether its handle is in safe mode through these.
static int multi_handles(Pool &p, const char *what, int ndev)
{
    if (ndev < 1 || (int)p.handle.size() < ndev) {
        gpx_set_error("%s: no handles for %d devices yet (make a multi-device call first)", what,
                      ndev);
        return -1;
    }
    for (int i = 0; i < ndev; ++i)
        if (!p.handle[i]) {
            gpx_set_error("%s: device %d has no handle yet", what, i);
            return -1;
        }
    return 0;
}

int gpx_multi_enable_timing(int ndev, int on)
{
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    GPX_TRY(multi_handles(p, "gpx_multi_enable_timing", ndev));
    for (int i = 0; i < ndev; ++i) GPX_TRY(gpx_enable_timing(p.handle[i], on));
    (void)hipSetDevice(0);
    return 0;
}

int gpx_multi_batch_info(int ndev, int64_t B_per_dev, int want_grad, double *dense_ms,
                         int64_t *members, int *plans)
{
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    GPX_TRY(multi_handles(p, "gpx_multi_batch_info", ndev));
    for (int i = 0; i < ndev; ++i) {
        double ms = 0.0;
        int64_t mem = 0;
        GPX_TRY(gpx_batch_timings(p.handle[i], &ms, &mem));
        if (dense_ms) dense_ms[i] = ms;
        if (members) members[i] = mem;
        if (plans) GPX_TRY(gpx_batch_plan(p.handle[i], B_per_dev, want_grad, plans + 4 * i));
    }
    (void)hipSetDevice(0);
    return 0;
}

int gpx_posterior_batch_multi(const gpx_kspec *k, const double *thetas, int64_t B,
                              const double *X, const double *y, int64_t n, int64_t d,
                              const double *Xs, int64_t m, int want_grad, int ndev, double *mu,
                              double *s2, double *dmu, double *ds2, int *info)
{
    if (!k || !thetas || !Xs || !mu || !s2 || B < 0 || m < 1 || ndev < 1 || (!X) != (!y) ||
        (X && (n < 1 || d < 1)) || (!dmu) != (!ds2)) {
        gpx_set_error("gpx_posterior_batch_multi: bad arguments");
        return -1;
    }
    const int nth = 1 + k->nhyper + 1;
    const bool grad = want_grad && dmu;
    int64_t dd = d;                                           // columns of Xs
    if (!X) {
        std::lock_guard<std::mutex> lock(pool().mu);
        dd = pool().res_d;
    }
    if (grad && dd < 1) {
        gpx_set_error("gpx_posterior_batch_multi: no resident data");
        return -1;
    }
    // per member: [mu (m) | s2 (m) | dmu (m d) | ds2 (m d) | info]
    const int64_t w64 = 2 * m + (grad ? 2 * m * dd : 0) + 1;
    if (w64 > (1 << 28)) {
        gpx_set_error("gpx_posterior_batch_multi: %lld doubles per member", (long long)w64);
        return -1;
    }
    const int width = (int)w64;
    std::vector<double> rows((size_t)B * width);
    GPX_TRY(multi_run("gpx_posterior_batch_multi", ndev, B, width, X, y, n, d, rows.data(),
                      [&](int, gpx_t *h, int64_t lo, int64_t cnt, double *out) -> int {
                          std::vector<double> a(cnt * m), v(cnt * m),
                              ga(grad ? cnt * m * dd : 0), gv(grad ? cnt * m * dd : 0);
                          std::vector<int> inf(cnt, 0);
                          const int r = gpx_posterior_batch(h, k, thetas + lo * nth, cnt, Xs, m,
                                                            a.data(), v.data(),
                                                            grad ? ga.data() : nullptr,
                                                            grad ? gv.data() : nullptr, inf.data());
                          for (int64_t b = 0; b < cnt; ++b) {
                              double *row = out + b * width;
                              memcpy(row, a.data() + b * m, m * 8);
                              memcpy(row + m, v.data() + b * m, m * 8);
                              if (grad) {
                                  memcpy(row + 2 * m, ga.data() + b * m * dd, m * dd * 8);
                                  memcpy(row + 2 * m + m * dd, gv.data() + b * m * dd, m * dd * 8);
                              }
                              row[width - 1] = (double)inf[b];
                          }
                          return r;
                      }));
    for (int64_t b = 0; b < B; ++b) {
        const double *row = rows.data() + b * width;
        memcpy(mu + b * m, row, m * 8);
        memcpy(s2 + b * m, row + m, m * 8);
        if (grad) {
            memcpy(dmu + b * m * dd, row + 2 * m, m * dd * 8);
            memcpy(ds2 + b * m * dd, row + 2 * m + m * dd, m * dd * 8);
        }
        if (info) info[b] = (int)row[width - 1];
    }
    return 0;
}

}  // extern "C"
