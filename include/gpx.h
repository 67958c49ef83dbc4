/*
 * gpx.h -- C ABI of libgpx.so, the MI355X-native (gfx950, HIP) exact-GP hot path
 * that replaces the NumPy/SciPy arithmetic of mwhoffman/pygp:
 *
 *   pygp/kernels/ (se, matern, periodic, _combo)  pairwise kernel evaluation
 *                                                  -> gpx_kernel_get/_grad
 *   pygp/inference/exact.py    ExactGP._update              -> gpx_exact_update
 *                              ExactGP.loglikelihood        -> gpx_exact_loglik
 *                              ExactGP._marg_posterior      -> gpx_exact_posterior
 *
 * pygp has no FFI of its own (pure Python, duck-typed Kernel / GP classes), so
 * these entry points are what a ctypes binding inside pygp would call; the
 * binding is shown in INTEGRATION.md and implemented in pygp_amd/_lib.py.
 *
 * Conventions
 *   - All host arrays are C-contiguous (row-major) float64 unless a dtype
 *     argument says otherwise; the caller owns them, the library copies in/out
 *     and keeps no host pointer after return.
 *   - Hyperparameters are in the reference's log-space layout
 *     (pygp/inference/_base.py:91-96): theta = [log sn | kernel hypers | mean].
 *   - Return value: 0 ok; >0 LAPACK-style info (1-based index of the first
 *     non-positive pivot, mirrors scipy.linalg.cholesky raising LinAlgError at
 *     pygp/inference/exact.py:54); <0 argument / HIP failure, text through
 *     gpx_last_error().
 *   - A handle owns one device, one HIP stream and all device memory. Calls on
 *     one handle must be serialised by the caller; distinct handles may be used
 *     from distinct threads (their diagonal-block launches are ordered on the
 *     device, one at a time) -- but one PROCESS per GPU: two processes on one
 *     device are not ordered against each other and may run into the 2-s wait
 *     bound of those launches (an error return, never a wrong result).
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_VERSION 100          /* major*10000 + minor*100 + patch */
#define GPX_MAX_DIM 32           /* input dimensions per kernel part */
#define GPX_MAX_PARTS 8          /* primitive kernels in the expanded sum of products */
#define GPX_MAX_HYPER (GPX_MAX_PARTS * (GPX_MAX_DIM + 2))

/* kernel families on the path (pygp/kernels/se.py, matern.py, periodic.py,
 * _combo.py SumKernel) */
enum gpx_kind {
    GPX_SE = 1,
    GPX_MATERN1 = 2,
    GPX_MATERN3 = 3,
    GPX_MATERN5 = 4,
    GPX_PERIODIC = 5,
    GPX_SUM = 6,
    GPX_RQ = 7,       /* rational quadratic (pygp/kernels/rq.py), hypers [sf, ell.., alpha] */
    GPX_PRODUCT = 8   /* product of kernels (_combo.py ProductKernel). Sums and products nest
                         freely: the tree is expanded into a sum of products of primitive
                         kernels (at most GPX_MAX_PARTS factors in all); a primitive that
                         the expansion repeats keeps ONE set of hyperparameters */
};

enum gpx_dtype { GPX_F64 = 0, GPX_F32 = 1 };

/* POD description of a kernel object. hyper = the kernel's get_hyper() vector
 * (log-space, reference order: [log sf, log ell...] for SE/Matern
 * (se.py:46-47, matern.py:62-63), [log sf, log ell, log p] for Periodic
 * (periodic.py:44-45)); for GPX_SUM / GPX_PRODUCT hyper is ignored and parts[]
 * holds the operands in order (_combo.py:90-98). */
typedef struct gpx_kspec {
    int32_t kind;                 /* enum gpx_kind */
    int32_t iso;                  /* 1: one shared lengthscale (se.py:33-36) */
    int32_t ndim;                 /* input dimensions */
    int32_t nhyper;               /* length of hyper (sum over parts for SUM) */
    const double *hyper;
    int32_t nparts;               /* GPX_SUM / GPX_PRODUCT only */
    const struct gpx_kspec *parts;
} gpx_kspec;

typedef struct gpx_ctx gpx_t;     /* opaque handle */

/* ---- library / device --------------------------------------------------- */
int gpx_version(void);
const char *gpx_last_error(void);            /* thread-local message */
int gpx_device_count(int *count);
int gpx_create(int device, gpx_t **out);
int gpx_destroy(gpx_t *h);
int gpx_synchronize(gpx_t *h);
/* Safe mode (own design; no counterpart in the reference): from now on this handle factors
 * diagonal blocks by recursion down to 128-tiles instead of task-queue launches whose
 * workgroups wait for each other. Those launches are ordered within a process; ANOTHER
 * process on the same GPU can starve them until their 2-s wait bound returns
 * "the panel kernel timed out ..." (<0). pygp_amd._lib can switch the handle to safe mode
 * on that error and repeat the call -- only if the caller asked for it
 * (Handle(auto_safe_mode=True) / GPX_AUTO_SAFE_MODE=1), with a RuntimeWarning that carries
 * the original error, and a counter on the handle; by default the error is raised. Same
 * tolerances, not the same bits: gpx_get_safe_mode and plan[3] of gpx_batch_plan say which
 * arithmetic a handle is on. */
int gpx_set_safe_mode(gpx_t *h, int on);
int gpx_get_safe_mode(gpx_t *h, int *on);   /* what the handle is in now (bench records, tests) */

/* ---- pairwise kernel evaluation (Kernel.get / Kernel.grad) -------------- */
/* K(X1, X2) -> out[n1*n2]; X2 == NULL means X2 = X1 (se.py:53-55,
 * matern.py:69-74, periodic.py:53-59, _combo.py:106-108). dtype selects the
 * element type of X1, X2 and out (GPX_F64 | GPX_F32). */
int gpx_kernel_get(gpx_t *h, const gpx_kspec *k, const void *X1, int64_t n1,
                   const void *X2, int64_t n2, int64_t d, int dtype, void *out);
/* all hyper-gradient slices -> out[nhyper*n1*n2] in hyper order (se.py:57-66,
 * matern.py:76-90, periodic.py:61-74, _combo.py:114-116). */
int gpx_kernel_grad(gpx_t *h, const gpx_kspec *k, const double *X1, int64_t n1,
                    const double *X2, int64_t n2, int64_t d, double *out);
/* input gradients of the kernel (RealKernel.gradx / grady: se.py:76-86,
 * matern.py:100-114, periodic.py:84-97, _real.py:96-100): out[n1*n2*d];
 * wrt = 1 -> d k / d x1, wrt = 2 -> d k / d x2. */
int gpx_kernel_gradx(gpx_t *h, const gpx_kspec *k, const double *X1, int64_t n1,
                     const double *X2, int64_t n2, int64_t d, int wrt, double *out);
/* device-resident variant of gpx_kernel_get for benchmarking the build alone:
 * X1 is taken from the handle's resident data (gpx_set_data), the result stays
 * in HBM; returns the kernel time in ms through *ms. */
int gpx_kernel_build_resident(gpx_t *h, const gpx_kspec *k, int dtype, int reps,
                              double *ms);

/* ---- ExactGP (pygp/inference/exact.py) ---------------------------------- */
/* upload the data once (GP.add_data, pygp/inference/_base.py:120-141) */
int gpx_set_data(gpx_t *h, const double *X, int64_t n, int64_t d,
                 const double *y);
/* ExactGP._update (exact.py:50-55): K + sn^2 I, R = chol, a = R^-T (y - mean).
 * *info receives the LAPACK-style pivot index (0 = ok). */
int gpx_exact_update(gpx_t *h, const gpx_kspec *k, double log_sn, double mean,
                     int *info);
/* ExactGP._updateinc (exact.py:57-62): append m observations to the data of the
 * current factorisation in O(n^2 m), in place: gpx_set_data reserves capacity for
 * at least 256 more points, so new 128-blocks open without a refactorisation (the
 * first append after an update completes R^-1 once). Returns -3 when an
 * incremental update is not possible (no current factor, or the capacity is
 * exhausted): the caller refactorises with gpx_set_data + gpx_exact_update, as
 * GP.add_data does on NotImplementedError (_base.py:132-141). If the extended
 * matrix is not positive definite (> 0) the handle keeps the old point count and
 * needs a new gpx_exact_update. */
int gpx_exact_append(gpx_t *h, const double *Xnew, const double *ynew, int64_t m,
                     int *info);
/* ExactGP.loglikelihood (exact.py:118-143) for the last update. dlZ == NULL ->
 * value only; else dlZ[1 + nhyper_kernel + 1] in order [sn, kernel..., mean]. */
int gpx_exact_loglik(gpx_t *h, double *lZ, double *dlZ);
/* One optimiser objective = set_hyper + loglikelihood(grad)
 * (pygp/learning/optimization.py:54-59) in a single call. */
int gpx_exact_eval(gpx_t *h, const gpx_kspec *k, double log_sn, double mean,
                   int want_grad, double *lZ, double *dlZ, int *info);
/* ExactGP._marg_posterior(grad=False) (exact.py:81-97) at m test points. */
int gpx_exact_posterior(gpx_t *h, const double *Xs, int64_t m, double *mu,
                        double *s2);
/* ExactGP._marg_posterior(grad=True) (exact.py:99-116): also d mu / d x and
 * d s2 / d x at the test points, dmu[m*d], ds2[m*d]. */
int gpx_exact_posterior_grad(gpx_t *h, const double *Xs, int64_t m, double *mu,
                             double *s2, double *dmu, double *ds2);
/* ExactGP._full_posterior (exact.py:64-79): mu[m] and the full covariance
 * Sigma[m][m] = K(Xs, Xs) - V^T V, V = R^-T K(X, Xs); 1 <= m <= 8192. GP.sample
 * (_base.py:143-178) draws from it. */
int gpx_exact_posterior_full(gpx_t *h, const double *Xs, int64_t m, double *mu, double *Sigma);
/* host copies of gp._R (n*n row-major upper, zero below the diagonal) and gp._a;
 * either may be NULL. n: the point count the caller sized R and a for; the call
 * fails when it is not the factor's. */
int gpx_exact_get_factor(gpx_t *h, int64_t n, double *R, double *a);
/* Batched hyperparameter evaluation on this handle's device: thetas[B*nth],
 * nth = 1 + k->nhyper + 1; kernel family/shape from k, hypers from thetas.
 * lZ[B]; dlZ[B*nth] or NULL; info[B] or NULL. (Sharding B over GPUs is done one
 * process per GPU above this call, see pygp_amd/batch.py.) The members run as groups in
 * lock-step, every kernel one launch over the group; member b returns the bits
 * gpx_exact_eval returns for thetas[b], whatever the batch. A member whose matrix is not
 * positive definite: lZ[b] = -inf, its dlZ NaN, its pivot in info[b]; the call still
 * returns 0. A non-finite theta is an error (< 0). The first call of a new shape allocates
 * the group workspaces (at most 40 % of the free device memory), kept until gpx_destroy. */
int gpx_loglik_batch(gpx_t *h, const gpx_kspec *k, const double *thetas,
                     int64_t B, int want_grad, double *lZ, double *dlZ,
                     int *info);
/* How gpx_loglik_batch / gpx_posterior_batch would run a batch of B thetas on the handle's
 * data right now (a function of the size, B and the free device memory): plan[0] = 0 one
 * context and stream per member (three in flight), 1 groups of members as one panel launch
 * each, 2 groups swept in lock-step, 3 groups as one launch with one workgroup per member
 * (many members at a small order); plan[1] members per group; plan[2] groups in flight;
 * plan[3] = 1 if the handle is in safe mode (gpx_set_safe_mode: another order of arithmetic),
 * else 0. For bench records and the first multi-GPU runs. */
int gpx_batch_plan(gpx_t *h, int64_t B, int want_grad, int *plan);
/* The same over the first ndev GPUs of the node, from one process and without
 * PyTorch: the B members are block-partitioned (gpx_batch_partition), every device
 * evaluates its block on a handle the library keeps for it (own host thread, X and
 * y replicated), and the per-member results are assembled with ONE ncclAllGather
 * on a single-process ncclCommInitAll communicator (RCCL over xGMI; librccl.so is
 * dlopen'ed on the first call with ndev > 1, ndev = 1 needs none). Replaces the
 * per-particle / per-sample Python loops pygp/meta/smc.py:102-126,
 * pygp/meta/mcmc.py:75-77. X == y == NULL: evaluate on the data a previous call
 * left resident on the devices. info[b] > 0: member b is not positive definite
 * (lZ[b] = -inf). ndev > device count -> -1. */
int gpx_loglik_batch_multi(const gpx_kspec *k, const double *thetas, int64_t B,
                           const double *X, const double *y, int64_t n, int64_t d,
                           int want_grad, int ndev, double *lZ, double *dlZ, int *info);
/* the same for gpx_posterior_batch: [m.posterior(X, grad) for m in samples]
 * (pygp/meta/mcmc.py:75-77, smc.py:128-130) dealt to ndev devices from one process, one
 * ncclAllGather. Arrays as in gpx_posterior_batch below; X, y as above (NULL: resident). */
int gpx_posterior_batch_multi(const gpx_kspec *k, const double *thetas, int64_t B,
                              const double *X, const double *y, int64_t n, int64_t d,
                              const double *Xs, int64_t m, int want_grad, int ndev, double *mu,
                              double *s2, double *dmu, double *ds2, int *info);
/* Audit of the multi-device entries, for benchmark records (own design): the handles the
 * library keeps for the first ndev devices exist after the first multi call with that many
 * devices (< 0 before). gpx_multi_enable_timing switches the HIP-event timing of their
 * groups on or off and clears the sums; gpx_multi_batch_info reports, per device, what
 * gpx_batch_timings (dense_ms[ndev], members[ndev]) and gpx_batch_plan for a block of
 * B_per_dev thetas (plans[ndev][4], plan[3] = safe mode) report for that device's handle.
 * Any output may be NULL. */
int gpx_multi_enable_timing(int ndev, int on);
int gpx_multi_batch_info(int ndev, int64_t B_per_dev, int want_grad, double *dense_ms,
                         int64_t *members, int *plans);
/* block [lo, hi) of `rank` when B members are dealt to `world` devices / ranks */
void gpx_batch_partition(int64_t B, int world, int rank, int64_t *lo, int64_t *hi);
/* Host halves of that gather (no device, no RCCL; exposed so that the packing rule is
 * unit-tested where there is no GPU). Every device contributes one slot of
 * gpx_multi_slot(B, ndev) = ceil(B / ndev) member rows of `width` doubles:
 * gpx_multi_pack copies a device's cnt <= slot rows into pack[slot * width] and pads the
 * rest with NaN (a device may have no member at all when B < ndev);
 * gpx_multi_scatter reads the gathered [ndev][slot][width] image back into
 * out[B][width] by the rule of gpx_batch_partition and never touches the padding. */
int64_t gpx_multi_slot(int64_t B, int ndev);
/* ranks of the RCCL communicator the last multi-device call gathered on (0: none yet,
 * e.g. only ndev = 1 calls so far) -- an audit value for benchmark records */
int gpx_multi_comm_size(void);
int gpx_multi_pack(const double *rows, int64_t cnt, int width, int64_t slot, double *pack);
int gpx_multi_scatter(const double *gathered, int64_t B, int ndev, int width, double *out);
/* [m.posterior(X, grad) for m in samples] of the meta-models (pygp/meta/mcmc.py:75-77,
 * pygp/meta/smc.py:128-130): B models on the resident data that differ only in their
 * hyperparameters, thetas[B][1 + nhyper + 1] = [log sn | kernel... | mean].
 * mu, s2: [B][m]; dmu, ds2: [B][m][d] or both NULL. info[b] > 0 (may be NULL): model b
 * is not positive definite, its rows are NaN. */
int gpx_posterior_batch(gpx_t *h, const gpx_kspec *k, const double *thetas, int64_t B,
                        const double *Xs, int64_t m, double *mu, double *s2, double *dmu,
                        double *ds2, int *info);

/* ---- instrumentation ---------------------------------------------------- */
/* per-stage GPU times (ms) of the last gpx_exact_eval / update+loglik measured
 * with HIP events on the handle's stream. names via gpx_timing_name(i). */
#define GPX_NTIMERS 10
int gpx_enable_timing(gpx_t *h, int on);
int gpx_get_timings(gpx_t *h, double *ms, int n);
/* batches that ran as groups (gpx_loglik_batch) since timing was last switched on: the sum of
 * the HIP-event times of the groups' factorisation (+ inverse) stages on their streams and
 * the members those groups held (two groups in flight overlap in time) */
int gpx_batch_timings(gpx_t *h, double *dense_ms, int64_t *members);
const char *gpx_timing_name(int i);

/* ---- dense building blocks (exposed for tests and micro-benchmarks) ------ */
/* C = alpha * op(A) * op(B) + beta * C on host row-major matrices; M, N, K are
 * padded internally. ta/tb: 0 = as stored, 1 = transposed. */
int gpx_la_gemm(gpx_t *h, int ta, int tb, int64_t M, int64_t N, int64_t K,
                double alpha, const double *A, int64_t lda, const double *B,
                int64_t ldb, double beta, double *C, int64_t ldc);
/* in: symmetric A (n*n, upper triangle used); out: R (upper, R^T R = A),
 * optionally Rinv (upper) and Ainv (symmetric, full) -- any may be NULL. */
int gpx_la_potrf(gpx_t *h, const double *A, int64_t n, double *R, double *Rinv,
                 double *Ainv, int *info);
/* device-resident timing of one n x n x n fp64 GEMM (TFLOP/s probe) */
int gpx_la_gemm_bench(gpx_t *h, int ta, int tb, int64_t n, int reps, double *ms);
/* same with the engine's structure knobs exposed (kernel experiments): flags =
 * GEMM_* bits of pygp_amd/csrc/gpx_internal.h, order/swizzle = tile walk,
 * tile = 0|64|128, waves = 0|4|8, same_ab: B aliases A */
int gpx_la_gemm_bench_ex(gpx_t *h, int ta, int tb, int64_t n, int flags, int order,
                         int swizzle, int tile, int waves, int same_ab, int reps,
                         double *ms);
/* device-resident timing of one M x N x K product (multiples of 128) with the
 * structure flags and a beta: the rank-K update shapes of the factorisation */
int gpx_la_gemm_bench_mnk(gpx_t *h, int ta, int tb, int64_t M, int64_t N, int64_t K, int flags,
                          double beta, int tile, int reps, double *ms);
/* device-resident timing of potrf (+potri if with_inverse) on a synthetic SPD
 * matrix of order n */
int gpx_la_potrf_bench(gpx_t *h, int64_t n, int with_inverse, int reps,
                       double *ms);

/* host-side self-check of the task graph of the diagonal-panel kernel for a block of T
 * 128-tiles (2..8; stream = 1: up to 32, a whole small matrix in one launch): schedule =
 * topological order of the counter dependencies, final counters, spine order; stream = 1
 * the round-2 graph, 0 the round-1 graph. No GPU. */
int gpx_panel_graph_check(int T, int workers, int stream, int *ntasks);
/* co-residency of a panel launch over nmem members on a device of ncu CUs (host only): every
 * workgroup of the launch holds a whole CU, and all spine workgroups plus at least one worker
 * must be resident together for the launch to make progress. *nspwg = spine workgroups per
 * member the launch would use (3 / 2 / 1 by members, fewer if they would not fit), *workers =
 * the shared pool; < 0: even one spine workgroup per member does not fit. */
int gpx_panel_grid_check(int nmem, int ncu, int *nspwg, int *workers);
/* the same for a wide panel launch: the block's T tiles plus E (0..8) tile columns to its
 * right, whose row-panel tiles and whose E x E diagonal block's update run inside the launch */
int gpx_panel_graph_check_wide(int T, int E, int workers, int *ntasks);
/* the same for a launch over a WHOLE matrix of T tiles (2..32) with the right-hand side of
 * the forward substitution as one more tile column */
int gpx_panel_graph_check_rhs(int T, int workers, int *ntasks);
/* ... and with ALL of R^-1 assembled inside the launch (an evaluation with gradients up to 32
 * tiles, round 5): the columns of the inverse as chunked sums beside the factorisation */
int gpx_panel_graph_check_full(int T, int workers, int *ntasks);
/* solo launches (one workgroup per member runs the member's whole task graph, an opt-in
 * arrangement of groups; GPX_SOLO_MAX_NP): the list is the graph in generation order -- checked
 * here to be a sequential order (every counter a task waits for reached, every producer a
 * follower polls finished) for T tiles [aug: + a right-hand side; full: the whole inverse
 * inside; value_only: without the inverse's tasks]. No GPU. */
int gpx_panel_solo_check(int T, int aug, int full, int value_only, int *ntasks);
/* host-side self-check of the lock-step sweep that factors the diagonal blocks of groups of
 * many members (T tiles, aug = 1: with a right-hand-side tile column): its phases and the
 * tile-engine updates between them, replayed against the counter thresholds of the panel
 * launch's task graph. No GPU. */
int gpx_sweep_check(int T, int aug);
/* the same for the sweep with DENSE row panels (round 5, the default): the tiles right of
 * (s, s+1) solved two workgroups a CU with the tile in registers, each applying the last
 * `depth` trailing updates of its tile itself (depth < 0: the library's rule for T tiles) */
int gpx_sweep_check_lite(int T, int aug, int depth);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
