// Internal declarations shared by the HIP translation units of libgpx.so.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/gpx.h"

#define GPX_TILE 128           // block-tile edge of the dense engine; every
                               // device matrix is padded to a multiple of it
#define GPX_BK 16
#define GPX_PANEL_MAX 1024    // largest diagonal block factored by one panel launch
#define GPX_PANEL_WHOLE_MAX 4096   // largest whole matrix factored by one panel launch
#define GPX_PANEL_WHOLE_DEFAULT 4096   // ... by default (GPX_PANEL_WHOLE)

// ---- error plumbing --------------------------------------------------------
void gpx_set_error(const char *fmt, ...);
#define GPX_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            gpx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                          hipGetErrorString(e_));                              \
            return -2;                                                         \
        }                                                                      \
    } while (0)
#define GPX_TRY(expr)                                                          \
    do {                                                                       \
        int r_ = (expr);                                                       \
        if (r_ < 0) return r_;                                                 \
    } while (0)

// ---- flattened kernel parameters (passed to kernels by value) -------------
// One primitive part of a (sum) kernel. scale[d] is the per-dimension divisor
// applied to the inputs exactly as the reference does (x / ell for SE,
// x / (ell / sqrt(d)) for Matern, 1 for Periodic).
struct KPart {
    int kind;            // gpx_kind (never GPX_SUM)
    int iso;
    int nhyper;
    int hoff;            // offset of this part's hypers in the kernel's vector
    int group;           // parts of one group multiply, groups add (sum of products)
    int dup;             // this primitive appeared in an earlier group too (a sum inside a
                         // product, expanded): its gradient slots accumulate
    double two_logsf;    // 2 * log sf
    double sf2;          // exp(2 log sf)
    double ell;          // Periodic: exp(log ell)
    double pi_over_p;    // Periodic: pi / exp(log p) (the fp32 build, tolerance 1e-5)
    double period;       // Periodic: exp(log p); fp64 code forms r * pi / p in the reference's order
    double alpha;        // RQ: exp(log alpha)
    double scale[GPX_MAX_DIM];
};
struct KParams {
    int nparts;
    int ndim;
    int nhyper;
    int nprod;           // parts that share their group with another part
    KPart part[GPX_MAX_PARTS];
};
int gpx_flatten_kspec(const gpx_kspec *k, int64_t d, KParams *out);

// ---- member-batched launches (round 4) ----------------------------------------
// gpx_loglik_batch / gpx_posterior_batch evaluate GROUPS of members in lock-step: every
// kernel of the sequence is ONE launch over all members of the group (blockIdx.z = member,
// or the member folded into the task queue of the panel kernel) instead of one launch
// sequence and one stream per member. The members share the data and the shapes and
// differ in their hyperparameters, which sit in device memory, one record per member;
// their matrices and vectors lie `mstride` / `vstride` elements apart.
struct MemberParams {
    KParams kp;
    double sn2;          // exp(2 log sn), added to the diagonal (gaussian.py:36-39)
    double mean;
    double prior;        // k(x, x): sum over groups of the product of sf^2 (Kernel.dget)
    double pad_;
};
struct MemberBatch {
    int count = 1;                           // members of the launch (1: not batched)
    const MemberParams *params = nullptr;    // device, [count] (null: kp by value)
    long long mstride = 0;                   // between the members' np x ld matrices
    long long vstride = 0;                   // between the members' vectors of np
};
int gpx_kspec_with_hyper(const gpx_kspec *k, const double *hyper,
                         std::vector<gpx_kspec> &store, gpx_kspec *out);

// ---- dense engine ----------------------------------------------------------
enum {
    GEMM_UPPER_ONLY = 1,   // skip C tiles entirely below the diagonal
    GEMM_KLO_M = 2,        // op(A)[m][k] == 0 for k <  m0       (start k at m0)
    GEMM_KHI_M = 4,        // op(A)[m][k] == 0 for k >= m0+TILE  (stop  k there)
    GEMM_KLO_N = 8,        // op(B)[k][n] == 0 for k <  n0
    GEMM_KHI_N = 16,       // op(B)[k][n] == 0 for k >= n0+TILE
    GEMM_KREV = 32,        // walk k from the top of the range downwards
};
struct GemmArgs {
    const double *A;
    const double *B;
    double *C;
    int lda, ldb, ldc;
    int M, N, K;           // multiples of GPX_TILE (K: multiple of GPX_BK)
    double alpha, beta;
    long long strideA, strideB, strideC;   // batch strides (elements)
    int batch;
    int flags;
    int tile;              // 0 = choose by grid size, 64 or 128 = force
    int order;             // tile walk, so that the longest k-ranges start first:
                           // 0 row-major, 1 row-major from the last tile row,
                           // 2 column-major from the last tile column,
                           // 3 column-major from the first tile column
    int swizzle;           // 1: XCD-aware 8x8 macro-tile walk when the grid allows
    int waves;             // 0 = default wave geometry, 4 or 8 = force
    int use_lists;         // 1: structured launches walk a sorted live-tile list
    const int *tiles;      // set by the launcher
    double *C2 = nullptr;  // C is a diagonal block of a matrix: its off-diagonal 128-tiles
                           // are read and written at C2 (same ldc) instead of C
    int slots = 0;         // workgroup slots of the stream the launch goes to (2 per CU
                           // the stream may use; 0 = the whole GPU): equal-k launches
                           // run whole rounds of 128-tiles and the rest as 64-tiles
    int kshift = 0;        // GEMM_KLO_*: the zero structure starts kshift columns in,
                           // op(A)[m][k] == 0 for k < m0 - kshift (a block column of a
                           // triangular matrix whose diagonal block sits kshift rows down)
    int beta0_from = -1;   // >= 0: tiles with n0 >= beta0_from take beta = 0 (a new block
                           // column of an accumulated matrix)
    int overlap = 0;       // 1: launches of other streams run beside this one (look-ahead):
                           // a partly filled last round is filled by them, and the
                           // whole-rounds + remainder split only adds a launch (round 3:
                           // 71.5 -> 70.8 ms per evaluation without it; batches, one stream
                           // per evaluation, keep it)
    long long strideC2 = 0;// batch stride of C2
    int kchunk = 0;        // > 0 (multiple of 64): split-K. Batch index z multiplies the
                           // SAME A and B over k in [z*kchunk, (z+1)*kchunk) only and
                           // writes its partial product to C + z*strideC
    int nsplit = 0;        // > 0 with kchunk: split-K for several MEMBERS in one launch.
                           // z = member * nsplit + chunk; the chunk cuts k as above and
                           // writes to C + chunk*strideC, the member moves A, B and C by
                           // mstrideA / mstrideB / mstrideC
    long long mstrideA = 0, mstrideB = 0, mstrideC = 0;
};
// C = alpha op(A) op(B) + beta C, ta/tb: 0 = stored [row][k] / [k][col].
int gpx_gemm(hipStream_t s, int ta, int tb, const GemmArgs &g);
// +1 / -1 around a region that keeps several evaluations in flight on different
// streams of `device` (selects the tile order of structured launches there)
void gpx_gemm_concurrency(int device, int delta);
int gpx_gemm_concurrent(int device);          // current count


struct DenseWs {           // device buffers of one factorisation, all np x np
    double *A = nullptr;   // K + sn2 I  ->  R (upper)
    double *W = nullptr;   // R^-1 (upper); diagonal leaves filled by potrf
    double *Kinv = nullptr;// (R^T R)^-1 upper; also temp of trtri
    int np = 0;            // padded order
    int ld = 0;            // leading dimension (np + pad: keeps rows off one HBM channel)
    int *info = nullptr;   // device int
    int *pctl = nullptr;   // control block of the panel kernel (zero between launches)
    // member-batched workspaces (round 4): `batch` members whose A / W / Kinv lie mstride
    // elements apart, info one int apart, control blocks pstride ints apart; every launch
    // of the factorisation covers all of them (no look-ahead streams then)
    int batch = 1;
    long long mstride = 0;
    int pstride = 0;
    // one panel launch over a whole matrix of at most GPX_PANEL_WHOLE_MAX (value-only):
    // decided ONCE by the caller (gpx_potrf_whole_ok) and honoured by gpx_potrf
    bool whole = false;
    // nothing will read W = R^-1 after this factorisation (value-only batch members whose
    // forward substitution rides along with a whole-matrix launch): lock-step sweeps then
    // skip the inverse; R and a do not depend on it
    bool no_inverse = false;
    // (with `whole`, more than one default block) the factorisation leaves ALL of W = R^-1
    // behind, not only the inverses of the 1024-blocks: the route of an evaluation with
    // gradients (gpx_grad_mode returned GPX_POTRF_R), which then skips gpx_trtri
    bool full_w = false;
    int *gate_total = nullptr;   // host: how often each of the two gate counters behind the
                                 // control block has been moved by launches enqueued so far
    // look-ahead of gpx_potrf (all null: everything on the caller's stream): a
    // high-priority stream for the diagonal blocks, a low-priority one for the left
    // half of the inverse tree, and GPX_LA_EVENTS events
    hipStream_t crit = nullptr, aux = nullptr;
    hipStream_t crit_only = nullptr;   // the reserved CUs and nothing else
    // the trailing updates run on `bulk`, a stream whose CU mask leaves a few CUs
    // (one or two per XCD) free: a 128-KB leaf workgroup of the next diagonal block
    // never finds room on a CU that holds two 72-KB GEMM workgroups, and would
    // otherwise wait for the whole update launch to drain
    hipStream_t bulk = nullptr;
    int bulk_slots = 0;    // workgroup slots of `bulk` (2 per unmasked CU)
    hipEvent_t *events = nullptr;
    // set by the caller of gpx_potrf (look-ahead only):
    // lead: event after which the first lead_rows rows of the input are in place (the
    // rest follows on the caller's stream): the first diagonal block starts on it while
    // the matrix is still being built
    hipEvent_t lead = nullptr;
    int lead_rows = 0;
    // defer_kinv: with GPX_POTRF_KINV, return to the caller's stream as soon as R and
    // W = R^-1 are complete; the last K^-1 update is still running on `aux` then and
    // gpx_potrf_join must be called before Kinv is read
    bool defer_kinv = false;
    // a whole-matrix launch (gpx_potrf_whole; round 4: also the single panel of a matrix of
    // 256 .. GPX_PANEL_MAX rows, gpx_potrf_rhs_ok) also solves R^T a = r for the right-hand side
    // the caller has put into column np of the staging matrix (rows 0 .. np-1; the 127
    // columns right of it zero): a comes back in column np of A
    bool aug_rhs = false;
};
#define GPX_MAX_BLOCKS 64     // diagonal blocks of the right-looking factorisation
#define GPX_LA_EVENTS (4 * GPX_MAX_BLOCKS + 4)
// Diagonal blocks of the right-looking factorisation of a matrix of padded order np:
// a first block of 1024 rows (it is factored with nothing to hide under), then blocks
// of nb rows (2048 above np = 8192: rank-2048 updates run at 69 TFLOP/s against 63 for
// rank-1024; 1024 below, where the diagonal blocks are the critical path). GPX_NB0 /
// GPX_NB override the two sizes, GPX_BLOCKS="1024,2048,3072,..." gives the list (the
// last size repeats).
// fills offs[0..count] (offs[0] = 0, offs[count] = np) and returns count
// full_inverse: the layout of a sweep that leaves the WHOLE of R^-1 behind
// (GPX_POTRF_KINV). Whatever runs after it may use any block partition (a diagonal block of
// R^-1 is the inverse of that diagonal block of R), so this one may differ from the
// default that a value-only factorisation and its later trtri / lauum must share: with
// GPX_SPLIT_LAST=1, for 4096 <= np < 8192 the last 1024-block is cut in two. The inverse
// column and K^-1 share of the last block are what nothing hides (a quarter of an
// evaluation at N = 4096). Worth 3 % there while the products ran on CU-masked streams;
// nothing since they run on every CU (2.72 ms either way), so off by default.
int gpx_block_layout(int np, int *offs, bool full_inverse = false);
struct GpxBlocks {
    int np, count;
    int offs[GPX_MAX_BLOCKS + 1];
    explicit GpxBlocks(int np_, bool full_inverse = false) : np(np_)
    {
        count = gpx_block_layout(np, offs, full_inverse);
    }
    int off(int k) const { return k <= 0 ? 0 : (k >= count ? np : offs[k]); }
    int len(int k) const { return off(k + 1) - off(k); }
};
// A -> R (upper). W receives R^-1 of every diagonal block of gpx_block_size rows (they
// are what the row-panel products multiply by). mode GPX_POTRF_R: nothing else;
// GPX_POTRF_W: the whole W = R^-1; GPX_POTRF_KINV: W and Kinv = W W^T = (R^T R)^-1
// (upper), built block column by block column beside the factorisation.
// Kinv is working storage: while a node waits for its row-panel step, its
// off-diagonal 128-tiles live in Kinv (the panel product reads them there and
// writes R into A, so nothing is ever copied); only diagonal tiles are updated
// in A. offdiag_staged: the caller already put the off-diagonal tiles of the
// input into Kinv (gpx_kbuild with out_offdiag); otherwise they are copied first.
enum { GPX_POTRF_R = 0, GPX_POTRF_W = 1, GPX_POTRF_KINV = 2 };
int gpx_potrf(hipStream_t s, const DenseWs &w, int mode, bool offdiag_staged);
// can gpx_potrf(w, mode) run as ONE panel launch over the whole matrix? A function of the
// matrix and the mode alone. The caller decides once (w.whole) and gpx_potrf honours it.
bool gpx_potrf_whole(const DenseWs &w, int mode);
// factorisation mode of an evaluation with gradients: GPX_POTRF_R (whole-matrix launch, then
// trtri + lauum) up to np = 4096, GPX_POTRF_KINV above (chol.hip)
int gpx_grad_mode(const DenseWs &w);
// ... and does that route assemble the whole of R^-1 inside the factorisation (w.full_w)?
bool gpx_grad_full_w(const DenseWs &w, int mode);
// may the caller set w.aug_rhs? (a whole-matrix launch, or a matrix that is one panel of at
// least two tiles, in any mode; ld must leave room for the tile column)
bool gpx_potrf_rhs_ok(const DenseWs &w, int mode);
// after a gpx_potrf with w.defer_kinv: make s wait for the last K^-1 update
// (a no-op when nothing was deferred)
// Test hook, GPX_TEST_JITTER=<seed>[:<max_us>] (tests/test_gpu_gp.py): a one-wave kernel
// that spins for a pseudo-random time (0 .. max_us, default 300; every other call) goes in
// front of every product, panel and build launch, on the stream of that launch. Stream
// order still holds, everything that is only ordered by luck moves: results must not
// change by a bit. No-op (one load) when unset.
int gpx_test_jitter(hipStream_t s);
int gpx_potrf_join(hipStream_t s, const DenseWs &w);
// the event a caller records into for w.lead
hipEvent_t gpx_potrf_lead_event(const DenseWs &w);
// complete W = R^-1 after a gpx_potrf(..., false)
int gpx_trtri(hipStream_t s, const DenseWs &w);
int gpx_lauum(hipStream_t s, const DenseWs &w);            // Kinv = W W^T
// X = R^-T B for B (np x m, ld ldb) in place, m multiple of GPX_TILE; T is a
// scratch of the same shape as B
// (w.batch members: their panels B and T lie pstride elements apart)
int gpx_trsm_rt(hipStream_t s, const DenseWs &w, double *B, double *T, int ldb, int m,
                long long pstride = 0);

// ---- vectors ---------------------------------------------------------------
// a = R^-T r (forward solve by 128-blocks with the leaf inverses in W);
// r is used as scratch and destroyed
size_t gpx_trsv_scratch(int np);      // doubles of `partial`
// (w.batch members: their vectors vstride, their scratch gpx_trsv_scratch(np) apart)
int gpx_trsv_rt(hipStream_t s, const DenseWs &w, bool w_complete, double *r_scratch,
                double *a, double *partial, long long vstride = 0);
// out = W v  (W upper triangular np x np)
int gpx_trmv_upper(hipStream_t s, const double *W, int ld, int np, const double *v,
                   double *out, int batch = 1, long long mstride = 0, long long vstride = 0);
// scalars[0] = sum_i a_i^2, scalars[1] = sum_i log R_ii (i < n), scalars[2] =
// sum_i alpha_i (if alpha); info != null: scalars[3] = the member's status word
int gpx_lz_terms(hipStream_t s, const double *R, int ld, int n, const double *a,
                 const double *alpha, double *scalars, int batch = 1, long long mstride = 0,
                 long long vstride = 0, int sstride = 0, const int *info = nullptr,
                 int astride = 1);
// r[i] = y[i] - mean (i < n), 0 for the padding
int gpx_residual(hipStream_t s, const double *y, double mean, int n, int np,
                 double *r);
// the members' residuals y - mean_b into r (may be null) and, with aug, into column np of
// their staging matrices (ld) as the right-hand side of a whole-matrix panel launch
int gpx_residual_members(hipStream_t s, const double *y, const MemberBatch &mb, int n, int np,
                         double *r, double *aug, int ld, int *info_zero = nullptr);
// the same for one model whose mean comes by value
int gpx_residual_rhs(hipStream_t s, const double *y, double mean, int n, int np, double *r,
                     double *aug, int ld);
// column `col` of the members' matrices into their vectors
int gpx_column_out(hipStream_t s, const double *A, int ld, int col, int np, double *out,
                   const MemberBatch &mb);
// mu[j] = mean + sum_i V[i][j] a[i];  s2[j] = prior - sum_i V[i][j]^2
size_t gpx_posterior_scratch(int m);
// V may come as nsplit partial sums, split_stride elements apart (split-K products)
int gpx_posterior_reduce(hipStream_t s, const double *V, int ldv, int np, int m,
                         const double *a, double mean, double prior, double *part,
                         double *mu, double *s2, int nsplit = 1, long long split_stride = 0,
                         const MemberBatch *mb = nullptr, long long vstride_panel = 0);
// out[e] = sum_s P[s * stride + e], e < count (count even): split-K partial products
// (batch members: their partial products sP, their outputs sout elements apart)
int gpx_sum_partials(hipStream_t s, const double *P, int nsplit, long long stride,
                     long long count, double *out, int batch = 1, long long sP = 0,
                     long long sout = 0);
// C[i][j] -= sum_s P[s * stride + i * cols + j] for the rows x cols block C (ld ldc)
int gpx_sub_partials(hipStream_t s, const double *P, int nsplit, long long stride, int rows,
                     int cols, double *C, int ldc);
// n x n host-shaped copies out of the padded np x np device matrices
int gpx_copy_upper(hipStream_t s, const double *A, int ld, int n, double *out);
int gpx_symmetrize(hipStream_t s, const double *A, int ld, int n, double *out);
int gpx_gemm_init();       // per-device kernel attributes (call after hipSetDevice)
int gpx_leaf2_init();
int gpx_panel_init();
int gpx_kmat_init();
// R and W = R^-1 of the diagonal block (off, n), 256 <= n <= gpx_panel_max(), in one
// launch (panel.hip); Kinv's block is scratch. extra > 0 (wide panel): the same launch
// also solves the row-panel columns [off + n, off + n + extra) -- R lands in A like any
// row panel -- and applies this block's update to the extra x extra diagonal block
// below them (diagonal tiles in A, the others in the staging area, as gpx_potrf keeps them)
// gate_need0 / 1: values the two gate counters of the workspace (gpx_panel_gates) must have
// reached before the launch touches the extra tiles of its own rows / of the next diagonal
// block: whoever applies the earlier updates to those tiles on another stream moves the
// gates behind them (0: nothing to wait for)
// rhs: the block is the whole matrix (off = 0, n = np <= gpx_panel_max()) and `extra` = 128
// is a right-hand-side tile column, as for the whole-matrix launches above GPX_PANEL_MAX
int gpx_panel(hipStream_t s, const DenseWs &w, int off, int n, int extra = 0,
              int gate_need0 = 0, int gate_need1 = 0, bool rhs = false);
int *gpx_panel_gates(const DenseWs &w);
// one phase of a lock-step sweep over the members of a batched workspace (panel.hip; the
// driver is sweep_block in chol.hip): phase 0 the leaf of tile (0,0), phase 1 + s the row
// panel of tile row s with the update and leaf of tile (s+1,s+1)
// a group of many members at a small order: one panel launch with one workgroup per member
// that runs the member's whole task graph (panel.hip) instead of the lock-step sweep
bool gpx_panel_solo(const DenseWs &w);
bool gpx_panel_solo_np(int np, int members);
int gpx_sweep_phase(hipStream_t s, const DenseWs &w, int off, int T, bool aug, int phase,
                    bool no_inverse, bool fused_only = false, bool presolved = false);
// the row-panel tiles (s, t >= t0) of every member as one dense launch: each tile takes its
// trailing updates kfirst .. s-1 itself, then its solve (panel.hip, sweep_xs_kernel)
bool gpx_sweep_lite();
int gpx_sweep_fold_depth(int T);
int gpx_sweep_right_max();
bool gpx_sweep_narrow();
int gpx_sweep_xs(hipStream_t st, const DenseWs &w, int off, int T, bool aug, int s, int t0,
                 int kfirst, int upd, int kfirst_rhs = -1);
int gpx_panel_max(int np);
bool gpx_panel_streaming();   // GPX_PANEL_STREAM != 0: the round-2 task graph        // block size used for a matrix of padded order np (0: none)
size_t gpx_panel_ctl_bytes();
// leaf factorisation of one 128x128 diagonal block: R (in place, upper, zeros
// below) and W = R^-1 (upper, zeros below) into Wblk. info (device int) gets
// goff + failing column + 1 if a pivot is not positive and *info == 0.
// (batch > 1: one workgroup per member, blocks mstride elements and info words one int apart)
int gpx_potrf_leaf2(hipStream_t s, double *Ablk, int lda, double *Wblk, int ldw,
                    int *info, int goff, int batch = 1, long long mstride = 0);

// ---- results of an evaluation from its device scalars ---------------------------
// sc[0] = sum a^2, sc[1] = sum log R_ii, sc[2] = sum alpha; acc[0] = tr(Q), acc[1 + h] =
// sum Q o dK_h. ONE definition for the single evaluation and the member-batched one: the
// members of a batch return the bits of the same evaluation on its own.
// Kernel.dget: k(x, x) = sum over groups of the product of sf^2 (se.py:68-69,
// _combo.py:110-112,128-131)
static inline double gpx_kernel_prior(const KParams &kp)
{
    double prior = 0.0, gprod = 0.0;
    for (int p = 0; p < kp.nparts; ++p) {
        if (p == 0 || kp.part[p].group != kp.part[p - 1].group) {
            prior += gprod;
            gprod = kp.part[p].sf2;
        } else {
            gprod *= kp.part[p].sf2;
        }
    }
    return prior + gprod;
}
static inline double gpx_assemble_lz(const double *sc, int n)
{
    return -0.5 * sc[0] - 0.5 * log(2 * M_PI) * n - sc[1];          // exact.py:119-121
}
static inline void gpx_assemble_dlz(const double *sc, const double *acc, double sn2, int nhyper,
                                    double *dlZ)
{
    dlZ[0] = -sn2 * acc[0];                                          // exact.py:134
    for (int i = 0; i < nhyper; ++i) dlZ[1 + i] = -0.5 * acc[1 + i]; // exact.py:137-138
    dlZ[1 + nhyper] = sc[2];                                         // exact.py:141
}

// ---- member-batched evaluation (group.hip) ---------------------------------------
struct GpxGroups;
// largest padded order evaluated in groups (GPX_GROUP_MAX_NP, default 32768; 0: never)
int gpx_groups_max_np();
int gpx_groups_min_big();
void gpx_groups_safe_mode(GpxGroups **state, int device, bool on);
// lZ (and dlZ) of B thetas on the device-resident data X (n x d), y: groups of members in
// lock-step, two groups in flight; *state is created on first use (per handle). Returns 1
// without having done anything when the batch is better served by the caller's own path
// (above np = 8192: fewer than four members fit a group)
int gpx_groups_loglik(GpxGroups **state, int device, const double *X, const double *y, int n,
                      int d, int np, const gpx_kspec *k, const double *thetas, int64_t B,
                      bool grad, double *lZ, double *dlZ, int *info);
// mu, s2 [and dmu, ds2] at the m test points Xs (host) of the B models theta: [B][m] ([B][m][d])
int gpx_groups_posterior(GpxGroups **state, int device, const double *X, const double *y, int n,
                         int d, int np, const gpx_kspec *k, const double *thetas, int64_t B,
                         const double *Xs, int64_t m, bool grads, double *mu, double *s2,
                         double *dmu, double *ds2, int *info);
// how gpx_groups_loglik would cut a batch of B thetas (g may be null): members per group
// (0: declined), groups in flight, and whether the groups are swept in lock-step
int gpx_groups_plan(const GpxGroups *g, int np, int64_t B, bool grad, int *members, int *inflight,
                    int *lockstep);
// stage timing of the groups (HIP events around each group's factorisation + inverse):
// switch, and the sums since the last reset
void gpx_groups_timing(GpxGroups **state, int device, bool on, bool reset);
void gpx_groups_get_timing(const GpxGroups *g, double *dense_ms, int64_t *members);
void gpx_groups_destroy(GpxGroups *g);

// ---- kernel-matrix kernels -------------------------------------------------
// generic pairwise evaluation: out[n1 x n2] (ld = ldo). If sym_upper, only
// tiles with col-tile >= row-tile are written. diag_add is added where
// (row == col) when X2 == X1 (sym). Rows/cols beyond n1/n2 up to the padded
// np1/np2 are written as identity (sym) or zero (cross). out_offdiag: off-diagonal
// 128-tiles go there (same ldo) instead of out -- the staging gpx_potrf expects.
// row0, rows (multiples of 64; with upper_only of 128; rows < 0: all): build rows
// [row0, row0 + rows) only.
template <typename T>
int gpx_kbuild(hipStream_t s, const KParams &kp, const T *X1, int n1, int np1,
               const T *X2, int n2, int np2, int d, T *out, long long ldo,
               bool sym, bool upper_only, double diag_add, T *out_offdiag = nullptr,
               int row0 = 0, int rows = -1, const MemberBatch *mb = nullptr);
int gpx_kgrad(hipStream_t s, const KParams &kp, const double *X1, int n1,
              const double *X2, int n2, int d, double *out);
// acc[0] = tr(Q), acc[1+h] = sum_ij Q_ij dK_h(i,j), Q = Kinv - alpha alpha^T,
// over the full symmetric matrix (computed from the upper triangle).
// partial: device scratch of at least gpx_trace_scratch(np) doubles.
size_t gpx_trace_scratch(int np);
int gpx_trace_grad(hipStream_t s, const KParams &kp, const double *X, int n,
                   int np, int d, const double *Kinv, int ld, const double *alpha,
                   double *partial, double *acc, const MemberBatch *mb = nullptr,
                   int astride = 0);

// d k / d x2 (sign = +1) or d k / d x1 (sign = -1): out[n1][n2][d]
int gpx_kgrady(hipStream_t s, const KParams &kp, const double *X1, int n1, const double *X2,
               int n2, int d, double sign, double *out);
// input gradients of the posterior mean / variance at m test points
// part: scratch of gpx_posterior_grad_scratch(n, m, d) doubles (row-chunk partials)
size_t gpx_posterior_grad_scratch(int n, int m, int d);
// (mb: the members' beta panels lie bstride, their partial sums
// gpx_posterior_grad_scratch(n, m, d) and their outputs m * d doubles apart)
int gpx_posterior_grad(hipStream_t s, const KParams &kp, const double *X, int n,
                       const double *Xs, int m, int d, const double *alpha,
                       const double *beta, int ldb, double *part, double *dmu, double *ds2,
                       const MemberBatch *mb = nullptr, long long bstride = 0);

// column strip [j0, j0+npc) of K + diag_add I for an appended block of observations
// (j0 a multiple of 128); out_offdiag: the off-diagonal 128-tiles go there instead
int gpx_kbuild_strip(hipStream_t s, const KParams &kp, const double *X, int n, int np,
                     int j0, int npc, int d, double *out, long long ldo, double diag_add,
                     double *out_offdiag = nullptr);
