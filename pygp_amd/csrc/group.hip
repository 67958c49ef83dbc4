// Member-batched evaluation of hyperparameter samples (round 4).
//
// The reference's particle / sample loops evaluate hundreds of thetas on the same few
// hundred to few thousand points, one after the other
// (/root/reference/pygp/meta/smc.py:102-126, /root/reference/pygp/meta/mcmc.py:75-77,
// /root/reference/pygp/learning/sampling.py:102-124,146). Rounds 1-3 gave every member of a
// batch its own context, stream and launch sequence and kept three in flight: below
// N = 4096 a member occupies a few dozen CUs for a chain of tile steps, three of them leave
// most of the GPU idle, a fourth collapsed on the runtime's hardware queues, and the host
// spent its time launching ~20 small kernels per member.
//
// Here a GROUP of members advances in lock-step on ONE stream: every kernel of the
// evaluation is one launch over all members of the group --
//   kernel build            blockIdx.z = member, hyperparameters from the member's record
//   diagonal blocks         fewer than 16 members: ONE panel launch whose task queue
//                           interleaves the members' task graphs (panel.hip), one control
//                           block per member; 16 or more: a lock-step sweep -- one tile row of
//                           the factorisation at a time over all members: a dense launch that
//                           updates and solves the row's tiles (two workgroups a CU), then the
//                           last diagonal update and the leaf (sweep_block in chol.hip)
//   products                the tile engine's batch dimension (blockIdx.z)
//   vector / trace kernels  blockIdx.z (or .x) = member
// -- and the members' few result doubles come back in one copy. Two groups are in flight
// (two streams) so that the chain-bound phase of one overlaps the products of the other
// and the host never waits for the GPU between groups. No per-member streams, no queue
// probing.
//
// A member takes exactly the arithmetic of the same evaluation on its own (gpx_exact_eval):
// the order of operations depends on (N, want_grad) only, so its bits do not depend on the
// group size, its slot in the group, or what else the device is doing
// (tests/test_gpu_groups.py). gpx_posterior_batch runs the same way: the group's update,
// then cross build, solve, reductions and input gradients one launch each over the group.

#include "gpx_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct Buf {
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

struct HostBuf {                                  // pinned
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need)
    {
        if (need <= bytes) return 0;
        if (p) GPX_HIP(hipHostFree(p));
        p = nullptr;
        bytes = 0;
        GPX_HIP(hipHostMalloc(&p, need));
        bytes = need;
        return 0;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

#define GROUP_SSTRIDE 4                           // doubles of scalars per member

struct Slot {                                     // one group in flight
    hipStream_t stream = nullptr;
    Buf A, W, Kinv, r, a, alpha, scalars, partial, gv_part, info, pctl, params;
    Buf Xs, Ks, KsT, mu, s2, post_part, split, gpart, dmu, ds2;     // posteriors
    HostBuf hparams, hres, hinfo;
    int cap = 0;                                  // members the buffers hold
    int np = 0, ld = 0;
    bool with_inverse = false;
    bool ctl_clean = false;
    bool no_panel = false;                        // safe mode (gpx_set_safe_mode): no task-queue
                                                  // launches, recursion down to the 128-tiles
    // the group in flight
    int count = 0;
    int64_t first = 0;
    bool grad = false;
    // stage timing (gpx_enable_timing on the handle): events around the factorisation
    // (+ inverse) of the group in flight
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool timed = false;
};

}  // namespace

struct GpxGroups {
    int device = 0;
    Slot slot[4];
    bool timing = false;
    bool no_panel = false;                        // (gpx_groups_safe_mode)
    double dense_ms = 0.0;                        // sum of the groups' factor(+inverse) stages
    int64_t dense_members = 0;                    // members those groups held
};

namespace {

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

int ld_for_group(int np)
{
    // room for the right-hand-side tile column of a whole-matrix launch, plus the padding
    // that keeps consecutive rows off one HBM channel (gpx_api.hip: ld_for)
    return np + 128 + 32;
}

int slot_reserve(Slot &s, int cap, int np, bool inverse)
{
    const int ld = ld_for_group(np);
    const size_t mat = (size_t)np * ld * 8, vec = (size_t)np * 8;
    const size_t pstride = (gpx_panel_ctl_bytes() / 4 + 63) / 64 * 64;
    if (cap > s.cap || np != s.np) s.ctl_clean = false;
    GPX_TRY(s.A.reserve(mat * cap));
    GPX_TRY(s.W.reserve(mat * cap));
    GPX_TRY(s.Kinv.reserve(mat * cap));
    GPX_TRY(s.r.reserve(vec * cap));
    GPX_TRY(s.a.reserve(vec * cap));
    GPX_TRY(s.gv_part.reserve(gpx_trsv_scratch(np) * 8 * cap));
    // the members' scalars [cap][GROUP_SSTRIDE] and, right behind them, their trace sums
    // [count][1 + nhyper]: the layout of the pinned result buffer, ONE copy a group
    GPX_TRY(s.scalars.reserve((size_t)(GROUP_SSTRIDE + GPX_MAX_HYPER + 2) * 8 * cap));
    GPX_TRY(s.info.reserve(sizeof(int) * cap));
    GPX_TRY(s.params.reserve(sizeof(MemberParams) * cap));
    GPX_TRY(s.hparams.reserve(sizeof(MemberParams) * cap));
    GPX_TRY(s.hres.reserve((size_t)(GROUP_SSTRIDE + GPX_MAX_HYPER + 2) * 8 * cap));
    GPX_TRY(s.hinfo.reserve(sizeof(int) * cap));
    if (s.pctl.bytes < pstride * 4 * cap) s.ctl_clean = false;
    GPX_TRY(s.pctl.reserve(pstride * 4 * cap));
    if (inverse) {
        GPX_TRY(s.alpha.reserve(vec * cap));
        GPX_TRY(s.partial.reserve(gpx_trace_scratch(np) * 8 * cap));
    }
    if (!s.ctl_clean) {
        // the panel kernel leaves its control blocks zero; a fresh allocation is not
        GPX_HIP(hipMemsetAsync(s.pctl.p, 0, s.pctl.bytes, s.stream));
        GPX_HIP(hipStreamSynchronize(s.stream));
        s.ctl_clean = true;
    }
    s.cap = std::max(s.cap, cap);
    s.np = np;
    s.ld = ld;
    return 0;
}

// members per group: enough of them to fill the GPU with the chain-bound phases (a member's
// panel keeps a few CUs busy), few enough that two groups fit into HBM many times over
int members_per_group(int np, bool grad)
{
    static const int forced = env_int("GPX_GROUP_MEMBERS", 0);
    if (forced > 0) return std::min(forced, 256);
    (void)grad;
    // (64 thetas at N = 8192, value-only / with gradients: 4 members per group 255 / 102
    // evals/s, 8: 276 / 106, 16: 306 / 110, 32: 320 / 113 -- from 16 on swept in lock-step)
    // (round 5, with the dense row panels of the sweep: 256 thetas value-only / with gradients,
    // 64 -> 128 members N = 1536 27.4k / 10.3k -> 28.5k / 10.6k, N = 2048 13.4k / 4.90k ->
    // 14.1k / 5.00k; 32 -> 64 members N = 3072 4.32k / 1.63k -> 4.70k / 1.70k, N = 4096
    // 2.05k / 745 -> 2.14k / 761)
    if (np <= 2048) return 128;
    if (np <= 4096) return 64;
    if (np <= 8192) return 32;
    // Above: the products are what takes the time and three contexts with look-ahead
    // streams of their own (rounds 2-3) already keep the GPU busy; one group of 6-8 members
    // in lock-step still gains 3 % (6 thetas at N = 16384: 14.35 -> 14.77 evals/s on the same
    // box, 3 + 3 or 2 + 2 + 2 in flight: 14.35 / 14.06) -- every product launch is six times
    // as long, its ramp and tail the same -- and needs no probing of hardware queues.
    return 8;
}

}  // namespace

// Fewest members of a group above np = 8192 (a shorter batch keeps one context per member).
// Rounds 3-4 kept two or three thetas on contexts with look-ahead streams of their own; with
// the contexts' pool on plain streams and panel launches ordered device-wide (end of round 4)
// a group of two or three is at least level everywhere and up to 48 % faster
// (tools/attic/r04_exp12.py, contexts -> group, evals/s: N = 9000 B = 2 120 -> 150 value-only,
// 61 -> 68 with gradients; B = 3 113 -> 167, 59 -> 72; N = 12000 B = 3 62 -> 84, 29 -> 33;
// N = 16384 B = 2 35.0 -> 36.8, 14.20 -> 14.02; B = 3 34.6 -> 38.7, 14.19 -> 14.33): 2.
int gpx_groups_min_big()
{
    static const int v = [] {
        const int e = env_int("GPX_GROUP_MIN_BIG", 2);
        return e < 2 ? 2 : e;
    }();
    return v;
}

int gpx_groups_max_np()
{
    static const int v = [] {
        // (32768 since the end of round 4: N = 20000 / 24000 / 32768, two or three thetas,
        // 1-8 % faster than the contexts and bit-equal to single evaluations,
        // tools/attic/r04_exp13.py, r04_exp14.py; 26 GB of workspaces per member at the top)
        const int e = env_int("GPX_GROUP_MAX_NP", 32768);
        return e < 0 ? 0 : e;
    }();
    return v;
}

void gpx_groups_destroy(GpxGroups *g)
{
    if (!g) return;
    static const bool dlog = getenv("GPX_DESTROY_LOG") != nullptr;
#define DLOG(...) do { if (dlog) { fprintf(stderr, "groups_destroy: " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
    (void)hipSetDevice(g->device);
    int si = 0;
    for (Slot &s : g->slot) {
        DLOG("slot %d sync", si);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        DLOG("slot %d buffers", si);
        Buf *bufs[] = {&s.A, &s.W, &s.Kinv, &s.r, &s.a, &s.alpha, &s.scalars, &s.partial,
                       &s.gv_part, &s.info, &s.pctl, &s.params, &s.Xs, &s.Ks, &s.KsT, &s.mu,
                       &s.s2, &s.post_part, &s.split, &s.gpart, &s.dmu, &s.ds2};
        for (Buf *b : bufs) b->release();
        DLOG("slot %d events", si);
        for (hipEvent_t e : s.ev)
            if (e) (void)hipEventDestroy(e);
        DLOG("slot %d host", si);
        s.hparams.release();
        s.hres.release();
        s.hinfo.release();
        DLOG("slot %d stream", si);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        ++si;
    }
    DLOG("done");
#undef DLOG
    delete g;
}

// what group_update leaves for the kernels that follow it on the slot's stream
struct GroupCtx {
    DenseWs w;
    MemberBatch mb;
    bool full_inverse = false;                    // W holds the whole R^-1
    bool a_in_column = false;                     // a = R^-T (y - m) still sits in column np of A
                                                  // (value-only members: nothing but the scalar
                                                  // terms reads it, no copy into the vector)
};

// members [first, first + count) on slot s: their parameter records, K + sn2 I, its
// factorisation in `mode` and a = R^-T (y - m); no host sync
static int group_update(Slot &s, const double *X, const double *y, int n, int d, int np,
                        const gpx_kspec *k, const double *thetas, int nth, int64_t first,
                        int count, int mode, bool lz_only, bool grad_eval, GroupCtx *out)
{
    MemberParams *hp = s.hparams.as<MemberParams>();
    std::vector<gpx_kspec> store;
    for (int i = 0; i < count; ++i) {
        const double *th = thetas + (first + i) * nth;
        for (int j = 0; j < nth; ++j)                    // (include/gpx.h: an error, < 0)
            if (!std::isfinite(th[j])) {
                gpx_set_error("non-finite hyperparameters (member %lld, component %d)",
                              (long long)(first + i), j);
                return -1;
            }
        gpx_kspec kb;
        GPX_TRY(gpx_kspec_with_hyper(k, th + 1, store, &kb));
        GPX_TRY(gpx_flatten_kspec(&kb, d, &hp[i].kp));
        hp[i].sn2 = exp(th[0] * 2);                      // gaussian.py:36-39
        hp[i].mean = th[nth - 1];
        hp[i].prior = gpx_kernel_prior(hp[i].kp);
        hp[i].pad_ = 0.0;
    }
    hipStream_t st = s.stream;
    GPX_HIP(hipMemcpyAsync(s.params.p, hp, sizeof(MemberParams) * count, hipMemcpyHostToDevice, st));

    const int ld = s.ld;
    DenseWs &w = out->w;
    w = DenseWs();
    w.A = s.A.as<double>();
    w.W = s.W.as<double>();
    w.Kinv = s.Kinv.as<double>();
    w.np = np;
    w.ld = ld;
    w.info = s.info.as<int>();
    w.pctl = s.no_panel ? nullptr : s.pctl.as<int>();
    w.batch = count;
    w.mstride = (long long)np * ld;
    w.pstride = (int)((gpx_panel_ctl_bytes() / 4 + 63) / 64 * 64);
    // (a lock-step sweep that assembles R^-1 forks its columns onto the slot's second stream)
    MemberBatch &mb = out->mb;
    mb = MemberBatch();
    mb.count = count;
    mb.params = s.params.as<MemberParams>();
    mb.mstride = w.mstride;
    mb.vstride = np;
    const KParams &kp0 = hp[0].kp;                       // the structure the members share

    // K + sn2 I: diagonal 128-tiles into A, the others into the staging area (exact.py:52)
    GPX_TRY(gpx_kbuild<double>(st, kp0, X, n, np, X, n, np, d, w.A, ld, true, true, 0.0, w.Kinv, 0,
                               -1, &mb));
    const bool whole = gpx_potrf_whole(w, mode);
    const bool aug = gpx_potrf_rhs_ok(w, mode);
    double *r = s.r.as<double>(), *a = s.a.as<double>();
    if (aug) {
        // a = R^-T (y - m) rides along with the factorisation as one more tile column (the
        // same launch clears the members' status words)
        GPX_TRY(gpx_residual_members(st, y, mb, n, np, nullptr, w.Kinv, ld, w.info));
        w.aug_rhs = true;
        w.no_inverse = lz_only && mode == GPX_POTRF_R;   // R and a are all that is read
    } else {
        GPX_HIP(hipMemsetAsync(s.info.p, 0, sizeof(int) * count, st));
    }
    w.whole = whole;
    w.full_w = grad_eval && gpx_grad_full_w(w, mode);   // (as the single evaluation, gpx_api.hip)
    if (s.timed) GPX_HIP(hipEventRecord(s.ev[0], st));
    GPX_TRY(gpx_potrf(st, w, mode, true));
    if (s.timed) GPX_HIP(hipEventRecord(s.ev[1], st));
    out->full_inverse = mode != GPX_POTRF_R || GpxBlocks(np).count == 1 || w.full_w;
    if (aug && w.no_inverse) {
        out->a_in_column = true;
    } else if (aug) {
        GPX_TRY(gpx_column_out(st, w.A, ld, np, np, a, mb));
    } else {
        GPX_TRY(gpx_residual_members(st, y, mb, n, np, r, nullptr, ld));
        GPX_TRY(gpx_trsv_rt(st, w, out->full_inverse, r, a, s.gv_part.as<double>(), mb.vstride));
    }
    return 0;
}

// enqueue the evaluation of members [first, first + count) on slot s; no host sync
static int group_enqueue(GpxGroups *g, Slot &s, const double *X, const double *y, int n, int d,
                         int np, const gpx_kspec *k, const double *thetas, int nth, int64_t first,
                         int count, bool grad)
{
    s.timed = g->timing;
    if (s.timed && !s.ev[0]) {
        GPX_HIP(hipEventCreate(&s.ev[0]));
        GPX_HIP(hipEventCreate(&s.ev[1]));
    }
    GroupCtx gc;
    s.no_panel = g->no_panel;
    // (with gradients: the mode the single evaluation takes at this order, gpx_grad_mode)
    int mode = GPX_POTRF_R;
    if (grad) {
        DenseWs wq;
        wq.np = np;
        wq.ld = s.ld;
        wq.pctl = s.no_panel ? nullptr : s.pctl.as<int>();
        mode = gpx_grad_mode(wq);
    }
    GPX_TRY(group_update(s, X, y, n, d, np, k, thetas, nth, first, count, mode, !grad, grad, &gc));
    const DenseWs &w = gc.w;
    const MemberBatch &mb = gc.mb;
    const KParams &kp0 = s.hparams.as<MemberParams>()[0].kp;
    hipStream_t st = s.stream;
    const int ld = s.ld;
    double *a = s.a.as<double>();
    double *scal = s.scalars.as<double>();
    double *hres = s.hres.as<double>();
    const int nacc = 1 + kp0.nhyper;
    if (grad) {
        // alpha = R^-1 a, the scalar terms, and the D + 2 trace terms (exact.py:127-141)
        double *alpha = s.alpha.as<double>();
        if (!gc.full_inverse) GPX_TRY(gpx_trtri(st, w));     // (value-only mode: complete R^-1)
        GPX_TRY(gpx_trmv_upper(st, w.W, ld, np, a, alpha, count, mb.mstride, mb.vstride));
        if (mode != GPX_POTRF_KINV) GPX_TRY(gpx_lauum(st, w));
        // (the status words ride in the scalars' fourth slot, the trace sums land right behind
        // the scalars: one result copy)
        GPX_TRY(gpx_lz_terms(st, w.A, ld, n, a, alpha, scal, count, mb.mstride, mb.vstride,
                             GROUP_SSTRIDE, w.info));
        GPX_TRY(gpx_trace_grad(st, kp0, X, n, np, d, w.Kinv, ld, alpha, s.partial.as<double>(),
                               scal + (size_t)GROUP_SSTRIDE * s.cap, &mb, nacc));
        GPX_HIP(hipMemcpyAsync(hres, scal, ((size_t)GROUP_SSTRIDE * s.cap + (size_t)nacc * count) * 8,
                               hipMemcpyDeviceToHost, st));
    } else {
        // a straight from the right-hand-side column where the factorisation left it
        if (gc.a_in_column)
            GPX_TRY(gpx_lz_terms(st, w.A, ld, n, w.A + np, nullptr, scal, count, mb.mstride,
                                 mb.mstride, GROUP_SSTRIDE, w.info, ld));
        else
            GPX_TRY(gpx_lz_terms(st, w.A, ld, n, a, nullptr, scal, count, mb.mstride, mb.vstride,
                                 GROUP_SSTRIDE, w.info));
        GPX_HIP(hipMemcpyAsync(hres, scal, (size_t)GROUP_SSTRIDE * 8 * count, hipMemcpyDeviceToHost, st));
    }
    s.count = count;
    s.first = first;
    s.grad = grad;
    (void)g;
    return 0;
}

// wait for the group on slot s and hand out its members' results
static int group_harvest(GpxGroups *g, Slot &s, int n, int nth, double *lZ, double *dlZ,
                         int *info)
{
    if (s.count == 0) return 0;
    GPX_HIP(hipStreamSynchronize(s.stream));
    if (s.timed) {
        float t = 0;
        GPX_HIP(hipEventElapsedTime(&t, s.ev[0], s.ev[1]));
        g->dense_ms += t;
        g->dense_members += s.count;
    }
    const double *hres = s.hres.as<double>();
    const MemberParams *hp = s.hparams.as<MemberParams>();
    const int nhyper = nth - 2, nacc = 1 + nhyper;
    int rc = 0;
    for (int i = 0; i < s.count; ++i) {
        const int64_t b = s.first + i;
        int inf = (int)hres[(size_t)GROUP_SSTRIDE * i + 3];  // (the status word: fourth scalar)
        if (inf < 0) {
            gpx_set_error("internal: the panel kernel timed out waiting for a dependency");
            rc = -1;
            inf = 0;
        }
        if (inf > n) inf = 0;                            // cannot happen: identity padding
        if (info) info[b] = inf;
        if (inf != 0) {                                  // not PD: -inf for a sampler
            lZ[b] = -INFINITY;
            if (s.grad && dlZ)
                for (int j = 0; j < nth; ++j) dlZ[b * nth + j] = NAN;
            continue;
        }
        const double *sc = hres + (size_t)GROUP_SSTRIDE * i;
        lZ[b] = gpx_assemble_lz(sc, n);
        if (s.grad && dlZ)
            gpx_assemble_dlz(sc, hres + (size_t)GROUP_SSTRIDE * s.cap + (size_t)nacc * i,
                             hp[i].sn2, nhyper, dlZ + b * nth);
    }
    s.count = 0;
    return rc;
}

// How a batch of B thetas is cut: *m members per group, *nslots groups in flight; *m = 0:
// declined, the caller's own path runs the batch (above np = 8192 with fewer than four
// members per group). g may be null (nothing allocated yet).
static int groups_plan(const GpxGroups *g, int np, int64_t B, bool grad, int *m_out,
                       int *nslots_out)
{
    static const int inflight_env = [] {
        const int v = env_int("GPX_GROUP_INFLIGHT", 0);
        return v < 1 || v > 4 ? 0 : v;
    }();
    // two groups in flight up to np = 8192; above, one (see members_per_group)
    const int inflight = inflight_env > 0 ? inflight_env : (np <= 8192 ? 2 : 1);
    int m = (int)std::min<int64_t>(members_per_group(np, grad), std::max<int64_t>(B, 1));
    {
        // three np x ld matrices per member (1.6 GB at np = 8192): the groups in flight take
        // at most 40 % of what is free on the device beyond what this handle's groups hold
        // already (other handles of the process -- the in-library multi-device path, the
        // rehearsal with faked devices -- and other processes share the same HBM)
        const double per = 3.0 * np * (double)ld_for_group(np) * 8 +
                           (grad ? gpx_trace_scratch(np) * 8.0 : 0.0);
        double held = 0.0;
        if (g)
            for (const Slot &s : g->slot) held += (double)(s.A.bytes + s.W.bytes + s.Kinv.bytes);
        const int64_t groups = std::max<int64_t>(1, std::min<int64_t>(inflight, (B + m - 1) / m));
        // the slots this cut would use already hold m members of this size: no query
        // (hipMemGetInfo is a driver call, and the 1.3 M evals/s tiny-dataset batches come
        // through here once per call)
        bool fits = g != nullptr;
        const size_t mat = (size_t)m * np * ld_for_group(np) * 8;
        for (int64_t i = 0; fits && i < groups; ++i) {
            const Slot &s = g->slot[i];
            fits = s.cap >= m && s.np == np && s.A.bytes >= mat && s.W.bytes >= mat &&
                   s.Kinv.bytes >= mat &&
                   (!grad || s.partial.bytes >= (size_t)gpx_trace_scratch(np) * 8 * m);
        }
        if (!fits) {
            size_t fr = 0, tot = 0;
            GPX_HIP(hipMemGetInfo(&fr, &tot));
            const double budget = 0.4 * ((double)fr + held);
            while (m > 1 && per * m * groups > budget) m = (m + 1) / 2;
        }
    }
    // large matrices: a group cut down to one member (memory) -- the caller keeps its own
    // path, one context with look-ahead per member (gpx_groups_min_big)
    if (np > 8192 && m < gpx_groups_min_big()) {
        *m_out = *nslots_out = 0;
        return 0;
    }
    // two groups in flight only when there is more than one group to run
    int nslots = (int)std::min<int64_t>(inflight, (B + m - 1) / m);
    if (nslots < 1) nslots = 1;
    // spread a short batch over the slots (B = 20 at m = 16: 10 + 10, not 16 + 4)
    if (nslots > 1 && B < (int64_t)m * nslots) m = (int)((B + nslots - 1) / nslots);
    *m_out = m;
    *nslots_out = nslots;
    return 0;
}

void gpx_groups_safe_mode(GpxGroups **state, int device, bool on)
{
    if (!*state) {
        GpxGroups *g = new (std::nothrow) GpxGroups();
        if (!g) return;
        g->device = device;
        *state = g;
    }
    (*state)->no_panel = on;
}

void gpx_groups_timing(GpxGroups **state, int device, bool on, bool reset)
{
    if (!*state) {
        GpxGroups *g = new (std::nothrow) GpxGroups();
        if (!g) return;
        g->device = device;
        *state = g;
    }
    (*state)->timing = on;
    if (reset) {
        (*state)->dense_ms = 0.0;
        (*state)->dense_members = 0;
    }
}

void gpx_groups_get_timing(const GpxGroups *g, double *dense_ms, int64_t *members)
{
    *dense_ms = g ? g->dense_ms : 0.0;
    *members = g ? g->dense_members : 0;
}

int gpx_groups_plan(const GpxGroups *g, int np, int64_t B, bool grad, int *members, int *inflight,
                    int *lockstep)
{
    GPX_TRY(groups_plan(g, np, B, grad, members, inflight));
    // (chol.hip: sweep_on)
    const int min_members = env_int("GPX_SWEEP_MIN_MEMBERS", 16);
    *lockstep = *members > 1 && min_members > 0 && *members >= min_members ? 1 : 0;
    // (panel.hip: one workgroup per member instead of the sweep, unless the handle is in safe mode)
    if (*lockstep && !(g && g->no_panel) && gpx_panel_solo_np(np, *members)) *lockstep = 2;
    return 0;
}

int gpx_groups_loglik(GpxGroups **state, int device, const double *X, const double *y, int n,
                      int d, int np, const gpx_kspec *k, const double *thetas, int64_t B,
                      bool grad, double *lZ, double *dlZ, int *info)
{
    if (!*state) {
        GpxGroups *g = new (std::nothrow) GpxGroups();
        if (!g) {
            gpx_set_error("groups: out of host memory");
            return -1;
        }
        g->device = device;
        *state = g;
    }
    GpxGroups *g = *state;
    const int nth = 1 + k->nhyper + 1;
    int m = 0, nslots = 0;
    GPX_TRY(groups_plan(g, np, B, grad, &m, &nslots));
    if (m == 0) return 1;                                // declined (see groups_plan)
    for (int i = 0; i < nslots; ++i) {
        Slot &s = g->slot[i];
        if (!s.stream) GPX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        GPX_TRY(slot_reserve(s, m, np, grad));
    }
    int rc = 0;
    int64_t done = 0;
    int gi = 0;
    for (; done < B && rc >= 0; ++gi) {
        Slot &s = g->slot[gi % nslots];
        rc = group_harvest(g, s, n, nth, lZ, dlZ, info);
        if (rc < 0) break;
        const int count = (int)std::min<int64_t>(m, B - done);
        rc = group_enqueue(g, s, X, y, n, d, np, k, thetas, nth, done, count, grad);
        done += count;
    }
    // the groups still in flight, oldest first (also on an error path: nothing may be left
    // running on buffers the next call reuses)
    for (int j = 0; j < nslots; ++j) {
        Slot &s = g->slot[(gi + j) % nslots];
        if (rc >= 0) {
            rc = group_harvest(g, s, n, nth, lZ, dlZ, info);
        } else {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            s.count = 0;
        }
    }
    return rc < 0 ? rc : 0;
}

// ---- posteriors of a batch of models (meta/mcmc.py:75-77, meta/smc.py:128-130) -----------
// C (np x mcp) = op(W) B for every member, W = R^-1 triangular (ta = 1: W^T, k < m0 + tile;
// ta = 0: W, k >= m0) -- gpx_api.hip's tri_product with the member dimension: with few
// columns the k range is cut into chunks whose partial products are summed in a fixed
// order, the cut depending on (np, mcp) only, as for a single model.
static int group_tri_product(Slot &s, const DenseWs &w, int count, int ta, const double *B,
                             double *C, int mcp)
{
    const int np = w.np;
    const long long pstride = (long long)np * mcp;
    GemmArgs g;
    g.A = w.W; g.B = B; g.C = C;
    g.lda = w.ld; g.ldb = mcp; g.ldc = mcp;
    g.M = np; g.N = mcp; g.K = np;
    g.alpha = 1.0; g.beta = 0.0;
    g.strideA = w.mstride; g.strideB = pstride; g.strideC = pstride;
    g.batch = count;
    g.flags = ta ? GEMM_KHI_M : GEMM_KLO_M;
    g.tile = 0; g.order = ta ? 1 : 0; g.swizzle = 0; g.waves = 0; g.use_lists = 1;
    g.tiles = nullptr;
    const long long tiles64 = (long long)(np / 64) * (mcp / 64);
    if (tiles64 > 1024 || np < 2048) return gpx_gemm(s.stream, ta, 0, g);
    const int kc = np >= 8192 ? 2048 : 1024;
    const int nsplit = (np + kc - 1) / kc;
    GPX_TRY(s.split.reserve((size_t)count * nsplit * pstride * 8));
    g.C = s.split.as<double>();
    g.kchunk = kc;
    g.nsplit = nsplit;
    g.batch = nsplit * count;
    g.strideA = g.strideB = 0;
    g.strideC = pstride;
    g.mstrideA = w.mstride;
    g.mstrideB = pstride;
    g.mstrideC = (long long)nsplit * pstride;
    g.tile = 64;
    GPX_TRY(gpx_gemm(s.stream, ta, 0, g));
    return gpx_sum_partials(s.stream, s.split.as<double>(), nsplit, pstride, pstride, C, count,
                            (long long)nsplit * pstride, pstride);
}

// test points per pass (gpx_api.hip: posterior_impl; the cut decides the split-K of the
// products, so it is the same function of np here)
static int posterior_chunk(int np) { return np <= 8192 ? 8192 : (np <= 16384 ? 4096 : 2048); }

static int group_posterior(Slot &s, const double *X, const double *y, int n, int d, int np,
                           const gpx_kspec *k, const double *thetas, int nth, int64_t first,
                           int count, const double *Xs, int64_t m, bool grads, double *mu,
                           double *s2, double *dmu, double *ds2, int *info)
{
    s.timed = false;
    GroupCtx gc;
    GPX_TRY(group_update(s, X, y, n, d, np, k, thetas, nth, first, count,
                         grads ? GPX_POTRF_W : GPX_POTRF_R, false, false, &gc));
    const DenseWs &w = gc.w;
    const MemberBatch &mb = gc.mb;
    const KParams &kp0 = s.hparams.as<MemberParams>()[0].kp;
    hipStream_t st = s.stream;
    const int ld = s.ld;
    GPX_HIP(hipMemcpyAsync(s.hinfo.p, s.info.p, sizeof(int) * count, hipMemcpyDeviceToHost, st));
    bool w_complete = gc.full_inverse;
    // V = R^-T K* is one triangle-aware product with W^T once W = R^-1 is complete, the
    // block substitution otherwise; completing W pays from np / 4 test points on (the rule
    // of a single model's first posterior call, gpx_api.hip)
    if (!w_complete && m >= np / 4) {
        GPX_TRY(gpx_trtri(st, w));
        w_complete = true;
    }
    double *a = s.a.as<double>();
    if (grads) {
        GPX_TRY(s.alpha.reserve((size_t)np * 8 * s.cap));
        GPX_TRY(gpx_trmv_upper(st, w.W, ld, np, a, s.alpha.as<double>(), count, mb.mstride,
                               mb.vstride));
    }
    const int CH = posterior_chunk(np);
    for (int64_t c0 = 0; c0 < m; c0 += CH) {
        const int mc = (int)std::min<int64_t>(CH, m - c0);
        const int mcp = (mc + GPX_TILE - 1) / GPX_TILE * GPX_TILE;
        const long long pstride = (long long)np * mcp;
        GPX_TRY(s.Xs.reserve((size_t)mc * d * 8));
        GPX_TRY(s.Ks.reserve((size_t)count * pstride * 8));
        GPX_TRY(s.KsT.reserve((size_t)count * pstride * 8));
        GPX_TRY(s.mu.reserve((size_t)count * mcp * 8));
        GPX_TRY(s.s2.reserve((size_t)count * mcp * 8));
        GPX_TRY(s.post_part.reserve((size_t)count * gpx_posterior_scratch(mcp) * 8));
        GPX_HIP(hipMemcpyAsync(s.Xs.p, Xs + c0 * d, (size_t)mc * d * 8, hipMemcpyHostToDevice, st));
        // K(X, Xs) of every member: np x mcp, zero outside n x mc (exact.py:87)
        MemberBatch mbp = mb;
        mbp.mstride = pstride;
        GPX_TRY(gpx_kbuild<double>(st, kp0, X, n, np, s.Xs.as<double>(), mc, mcp, d,
                                   s.Ks.as<double>(), mcp, false, false, 0.0, nullptr, 0, -1, &mbp));
        double *V = s.Ks.as<double>();
        if (w_complete) {                                // RK = R^-T K (exact.py:88)
            GPX_TRY(group_tri_product(s, w, count, 1, s.Ks.as<double>(), s.KsT.as<double>(), mcp));
            V = s.KsT.as<double>();
        } else {
            GPX_TRY(gpx_trsm_rt(st, w, s.Ks.as<double>(), s.KsT.as<double>(), mcp, mcp, pstride));
        }
        GPX_TRY(gpx_posterior_reduce(st, V, mcp, np, mcp, a, 0.0, 0.0, s.post_part.as<double>(),
                                     s.mu.as<double>(), s.s2.as<double>(), 1, 0, &mb, pstride));
        if (grads) {
            // beta = W V: W upper -> k >= row tile
            double *beta = (V == s.Ks.as<double>()) ? s.KsT.as<double>() : s.Ks.as<double>();
            GPX_TRY(group_tri_product(s, w, count, 0, V, beta, mcp));
            GPX_TRY(s.dmu.reserve((size_t)count * mc * d * 8));
            GPX_TRY(s.ds2.reserve((size_t)count * mc * d * 8));
            GPX_TRY(s.gpart.reserve((size_t)count * gpx_posterior_grad_scratch(n, mc, d) * 8));
            GPX_TRY(gpx_posterior_grad(st, kp0, X, n, s.Xs.as<double>(), mc, d,
                                       s.alpha.as<double>(), beta, mcp, s.gpart.as<double>(),
                                       s.dmu.as<double>(), s.ds2.as<double>(), &mb, pstride));
            GPX_HIP(hipMemcpy2DAsync(dmu + (first * m + c0) * d, (size_t)m * d * 8, s.dmu.p,
                                     (size_t)mc * d * 8, (size_t)mc * d * 8, count,
                                     hipMemcpyDeviceToHost, st));
            GPX_HIP(hipMemcpy2DAsync(ds2 + (first * m + c0) * d, (size_t)m * d * 8, s.ds2.p,
                                     (size_t)mc * d * 8, (size_t)mc * d * 8, count,
                                     hipMemcpyDeviceToHost, st));
        }
        GPX_HIP(hipMemcpy2DAsync(mu + first * m + c0, (size_t)m * 8, s.mu.p, (size_t)mcp * 8,
                                 (size_t)mc * 8, count, hipMemcpyDeviceToHost, st));
        GPX_HIP(hipMemcpy2DAsync(s2 + first * m + c0, (size_t)m * 8, s.s2.p, (size_t)mcp * 8,
                                 (size_t)mc * 8, count, hipMemcpyDeviceToHost, st));
        // the next pass reuses the panels (and the caller's arrays are pageable memory)
        GPX_HIP(hipStreamSynchronize(st));
    }
    GPX_HIP(hipStreamSynchronize(st));
    int rc = 0;
    const int *hinfo = s.hinfo.as<int>();
    for (int i = 0; i < count; ++i) {
        const int64_t b = first + i;
        int inf = hinfo[i];
        if (inf < 0) {
            gpx_set_error("internal: the panel kernel timed out waiting for a dependency");
            rc = -1;
            inf = 0;
        }
        if (inf > n) inf = 0;
        if (info) info[b] = inf;
        if (inf > 0) {                                   // not PD: this model has no posterior
            for (int64_t j = 0; j < m; ++j) mu[b * m + j] = s2[b * m + j] = NAN;
            if (grads)
                for (int64_t j = 0; j < m * d; ++j) dmu[b * m * d + j] = ds2[b * m * d + j] = NAN;
        }
    }
    return rc;
}

int gpx_groups_posterior(GpxGroups **state, int device, const double *X, const double *y, int n,
                         int d, int np, const gpx_kspec *k, const double *thetas, int64_t B,
                         const double *Xs, int64_t m, bool grads, double *mu, double *s2,
                         double *dmu, double *ds2, int *info)
{
    if (!*state) {
        GpxGroups *g = new (std::nothrow) GpxGroups();
        if (!g) {
            gpx_set_error("groups: out of host memory");
            return -1;
        }
        g->device = device;
        *state = g;
    }
    GpxGroups *g = *state;
    const int nth = 1 + k->nhyper + 1;
    // One group at a time (the host waits for every pass over the test points): as many
    // members as the factorisation workspaces AND the two np x mcp panels per member allow.
    int members = (int)std::min<int64_t>(members_per_group(np, grads), std::max<int64_t>(B, 1));
    {
        const int mcp = (int)std::min<int64_t>(posterior_chunk(np), (m + GPX_TILE - 1) / GPX_TILE * GPX_TILE);
        const double per = 3.0 * np * (double)ld_for_group(np) * 8 + 3.0 * np * (double)std::max(mcp, GPX_TILE) * 8;
        size_t fr = 0, tot = 0;
        GPX_HIP(hipMemGetInfo(&fr, &tot));
        double held = 0.0;
        for (const Slot &s : g->slot)
            held += (double)(s.A.bytes + s.W.bytes + s.Kinv.bytes + s.Ks.bytes + s.KsT.bytes);
        const double budget = 0.4 * ((double)fr + held);
        while (members > 1 && per * members > budget) members = (members + 1) / 2;
    }
    if (np > 8192 && members < gpx_groups_min_big()) return 1;   // (as gpx_groups_loglik)
    Slot &s = g->slot[0];
    if (!s.stream) GPX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    GPX_TRY(slot_reserve(s, members, np, false));
    int rc = 0;
    s.no_panel = g->no_panel;
    for (int64_t done = 0; done < B && rc >= 0; done += members) {
        const int count = (int)std::min<int64_t>(members, B - done);
        rc = group_posterior(s, X, y, n, d, np, k, thetas, nth, done, count, Xs, m, grads, mu, s2,
                             dmu, ds2, info);
    }
    if (rc < 0) (void)hipStreamSynchronize(s.stream);
    return rc < 0 ? rc : 0;
}
