// Blocked Cholesky, triangular inverse and symmetric inverse on the fp64 tile
// engine (gemm_f64.hip). Everything works on the UPPER factor R of the
// reference (R^T R = K + sn2 I, scipy.linalg.cholesky at
// /root/reference/pygp/inference/exact.py:54) in row-major storage.
//
//   potrf  right-looking over diagonal blocks (1024 rows first, then 2048; 1024
//          throughout up to np = 8192) with look-ahead. Block k is factored AND
//          inverted (F_k: the recursion below over one-launch 1024-panels), its row
//          panel is ONE triangle-aware MFMA GEMM R[k, k+1:] = W_kk^T A[k, k+1:] (no
//          substitution anywhere), then the next diagonal block gets its update
//          first and F_k+1 starts on a high-priority stream while `bulk` is still
//          busy with the rest of step k: the latency-bound chain of diagonal blocks
//          hides under the MFMA-bound products of ONE evaluation (the optimize()
//          pattern). Default: every stream over every CU, the two small products
//          between F_k and F_k+1 on the high-priority stream itself (fast chain);
//          GPX_RESERVE_CUS=32: the CU partition of the first half of round 2
//          (products on CU-masked streams, diagonal blocks on the reserved CUs).
//          With GPX_POTRF_W / _KINV the sweep also builds R^-1 and (R^T R)^-1 block
//          column by block column (inverse_column, on a third stream): the
//          shrinking trailing update and the growing inverse work add up to about
//          the same amount of products in every step.
//   inside a diagonal block: recursive -- factor the leading half AND invert its
//          factor, R12 = W11^T A12, SYRK the trailing half, recurse, extend the
//          inverse W12 = -W11 (R12 W22). Blocks of <= 1024 rows are one task-queue
//          launch (panel.hip); the 128x128 leaves are factored AND inverted by one
//          workgroup (leaf.hip).
//   trtri  W = R^-1 from the diagonal-block inverses by the same two products per
//          node of a binary tree over the blocks (after a value-only potrf).
//   lauum  Kinv = W W^T, upper tiles only, one launch with per-tile k ranges (ditto).
//
// potrf + trtri + lauum = N^3 flops, against the 7N^3/3 of the reference's
// cho_solve(R, eye(N)) route (exact.py:129).

#include "gpx_internal.h"

#define LB GPX_TILE                    // leaf order

// ---- environment knobs (developer experiments) --------------------------------
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include <mutex>
#include <string>
static int env_int(const char *name, int dflt)
{
    static std::map<std::string, int> cache;
    static std::mutex mu;                       // handles may live on several threads
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(name);
    if (it != cache.end()) return it->second;
    const char *e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    cache[name] = v;
    return v;
}


// ---- block copy (row panel -> scratch before its out-of-place multiply) ------
__global__ __launch_bounds__(256) void copy_block_kernel(const double *__restrict__ src,
                                                         double *__restrict__ dst, int ld,
                                                         int rows, int cols, long long stride)
{
    src += (long long)blockIdx.z * stride;                  // blockIdx.z = member
    dst += (long long)blockIdx.z * stride;
    const int c2 = blockIdx.x * 256 + threadIdx.x;          // double2 column index
    if (2 * c2 >= cols) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        reinterpret_cast<double2 *>(dst + (size_t)r * ld)[c2] =
            reinterpret_cast<const double2 *>(src + (size_t)r * ld)[c2];
}

static int copy_block(hipStream_t s, const double *src, double *dst, int ld, int rows,
                      int cols, int batch = 1, long long stride = 0)
{
    dim3 grid((cols / 2 + 255) / 256, rows < 1024 ? rows : 1024, batch);
    hipLaunchKernelGGL(copy_block_kernel, grid, dim3(256), 0, s, src, dst, ld, rows, cols,
                       stride);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- wide panels -----------------------------------------------------------------
// GPX_PANEL_WIDE (default 0: built, parity-green, SLOWER): a diagonal block of <= 1024
// rows whose right neighbour has <= 1024 rows is factored by a WIDE panel launch
// (panel.hip) that also solves the row-panel block R[k, k+1] and applies update k to block
// (k+1, k+1), so that the two dependent launches between two panels on the chain of
// diagonal blocks (N = 4096: 3 x ~0.17 of 2.7 ms) disappear. 1: the top-level chain (all
// blocks 1024: np <= 8192); 2: also inside 2048-blocks (first half -> second half).
// Measured (round 3, tools/r03_exp13.sh): N = 4096 2.67 -> 2.91 ms, N = 8192 11.5 -> 12.7,
// N = 16384 (level 2) 71 -> 78: the 2e9 flop moved into the launch run as 64 x 64 x 128
// product tasks of lone workgroups (0.11 TFLOP/s each, 19 ms of workgroup time per
// panel), an order of magnitude less efficient than the two tile-engine launches they
// replace, and they crowd the tasks the chain is waiting for.
// The extra tiles of a top-level wide panel still receive update k-1 from the trailing
// launches of step k-1 on another stream while the panel is already running: its tasks on
// those tiles wait for two gate counters of the workspace, moved by one-thread kernels
// behind those launches (gate 0: row k's tiles, gate 1: block (k+1, k+1)).
__global__ void gate_bump_kernel(int *gate)
{
    __hip_atomic_fetch_add(gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int wide_level();

static int gate_bump(hipStream_t s, const DenseWs &w, int which)
{
    // only with wide panels: even a one-thread kernel waits ~45 us for a wave slot between
    // two product launches (every SIMD holds four 128-VGPR waves), 0.6 ms per evaluation
    if (!w.gate_total || !w.pctl || wide_level() < 1) return 0;
    hipLaunchKernelGGL(gate_bump_kernel, dim3(1), dim3(1), 0, s, gpx_panel_gates(w) + which);
    GPX_HIP(hipGetLastError());
    ++w.gate_total[which];
    return 0;
}

static int wide_level()
{
    static const int v = env_int("GPX_PANEL_WIDE", 0);
    static const int graph = env_int("GPX_PANEL_STREAM", 1);   // the round-1 graph has no XS tasks
    return graph ? v : 0;
}

// block (off, n) followed by `next` rows: can one wide launch take both?
static bool wide_ok(const DenseWs &w, int n, int next, int level)
{
    const int pm = gpx_panel_max(w.np);
    return wide_level() >= level && w.pctl && w.gate_total && n >= 2 * LB && n <= pm &&
           next >= LB && next <= GPX_PANEL_MAX;
}

// ---- recursion ---------------------------------------------------------------
static inline int split(int n) { return (n / LB / 2) * LB; }   // leading half

// set for the duration of a gpx_potrf call that runs with look-ahead (GemmArgs::overlap)
static thread_local int tl_overlap = 0;
// set for the duration of a driver call on a member-batched workspace: every product is one
// launch over all members (blockIdx.z), whose matrices lie tl_mstride elements apart
static thread_local int tl_batch = 1;
static thread_local long long tl_mstride = 0;
struct BatchScope {
    int prev_batch;
    long long prev_stride;
    explicit BatchScope(const DenseWs &w) : prev_batch(tl_batch), prev_stride(tl_mstride)
    {
        tl_batch = w.batch > 1 ? w.batch : 1;
        tl_mstride = w.batch > 1 ? w.mstride : 0;
    }
    ~BatchScope()
    {
        tl_batch = prev_batch;
        tl_mstride = prev_stride;
    }
};

static GemmArgs mk(const double *A, int lda, const double *B, int ldb, double *C,
                   int ldc, int M, int N, int K, double alpha, double beta, int flags)
{
    GemmArgs g;
    g.overlap = tl_overlap;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    // (every operand of the drivers below lives in one of the members' np x ld matrices)
    g.strideA = g.strideB = g.strideC = g.strideC2 = tl_mstride;
    g.batch = tl_batch;
    g.flags = flags;
    g.tile = 0;
    g.order = 0;
    g.swizzle = env_int("GPX_SWIZZLE", 0);
    g.waves = 0;
    g.use_lists = env_int("GPX_TILE_LISTS", 1);
    g.tiles = nullptr;
    return g;
}

// C -= P^T P on the upper tiles of the n x n block C, P = k x n (the engine runs whole
// rounds of 128-tiles and the remainder as 64-tiles)
static int syrk_upper(hipStream_t s, const double *P, int ldp, double *C, int ldc, int n,
                      int k, double *C2, int slots = 0)
{
    GemmArgs g = mk(P, ldp, P, ldp, C, ldc, n, n, k, -1.0, 1.0, GEMM_UPPER_ONLY);
    g.C2 = C2;
    g.slots = slots;
    return gpx_gemm(s, 1, 0, g);
}

// W12 = -W11 (R12 W22) for the node (off, n) cut after n1 rows; Kinv's (1,2) block is
// scratch
static int extend_inverse(hipStream_t s, const DenseWs &w, int off, int n, int n1)
{
    const int ld = w.ld;
    const int n2 = n - n1;
    const size_t o11 = (size_t)off * ld + off, o12 = o11 + n1,
                 o22 = (size_t)(off + n1) * ld + off + n1;
    // T = R12 W22 : op(B) = W22 upper -> k <= column tile
    {
        GemmArgs g = mk(w.A + o12, ld, w.W + o22, ld, w.Kinv + o12, ld, n1, n2, n2, 1.0,
                        0.0, GEMM_KHI_N);
        g.order = env_int("GPX_ORD_T", 2);
        GPX_TRY(gpx_gemm(s, 0, 0, g));
    }
    // W12 = -W11 T : op(A) = W11 upper -> k >= row tile
    return gpx_gemm(s, 0, 0,
                    mk(w.W + o11, ld, w.Kinv + o12, ld, w.W + o12, ld, n1, n2, n1, -1.0,
                       0.0, GEMM_KLO_M | (env_int("GPX_KREV", 0) ? GEMM_KREV : 0)));
}

// ---- lock-step sweep of a block of tiles over all members of a batched workspace ------
// The arithmetic of the panel launch (panel.hip) for the block (off, n = 128 T) [and, with
// aug, a right-hand-side tile column right of a WHOLE matrix], run one tile row at a time over
// every member, as plain launches. Round 5 (GPX_SWEEP_LITE=0: round 4's phases, below):
//   F(0)
//   for s = 0 .. T-1:
//     [s > depth]  steps 0 .. kf-1 (kf = s - depth) of row s and of the next diagonal tile:
//                  one product each on the tile engine, K = 128 kf
//     dense launch (gpx_sweep_xs): every tile (s, t), t > s, takes its steps kf .. s-1 and its
//                  solve R_st = R_ss^-T X; the diagonal tile (s+1, s+1) its steps kf .. s-1
//     fused tasks (gpx_sweep_phase, presolved): A[s+1, s+1] -= R[s, s+1]^T R[s, s+1] from
//                  zero-based sums, as in the panel launch, then the leaf F(s+1)
// Round 4: the tile tasks of a phase as one launch of whole-CU workgroups, every trailing
// update a product of the tile engine, applied left-looking right before the row's phase:
//   for s:  [s >= 1]  X[s, s+1:] -= R[:s, s]^T R[:s, s+1:];  A[s+1, s+1] -= R[:s, s+1]^T R[:s, s+1]
//           X(s): R[s, t] = R_ss^-T X[s, t] for t > s; for t = s+1 followed by the last
//                 diagonal update and F(s+1)
// Either way a tile accumulates its updates in the order of the panel launch's step-by-step
// tasks: the same bits.
//   inverse (inside groups of 8 tiles, as the panel launch assembles it):
//   for s:  T = W[i0:s, i0:s] R[i0:s, s];  W[i0:s, s] = -T W_ss
static bool sweep_on(const DenseWs &w)
{
    // members from which a workspace is swept in lock-step instead of by one panel launch
    // with the members' task graphs interleaved (fewer members: the chain of one member is
    // what takes the time, and the panel launch overlaps its steps)
    static const int min_members = env_int("GPX_SWEEP_MIN_MEMBERS", 16);
    // (many members at a small order: one workgroup per member instead, gpx_panel_solo)
    return w.batch > 1 && min_members > 0 && w.batch >= min_members && gpx_panel_streaming() &&
           !gpx_panel_solo(w);
}

static int sweep_block(hipStream_t s, const DenseWs &w, int off, int n, bool aug, bool inverse)
{
    const int ld = w.ld, T = n / LB, TW = T + (aug ? 1 : 0);
    if (n % LB || T < 1 || (aug && (off != 0 || n != w.np || ld < n + LB))) {
        gpx_set_error("sweep: bad block (order %d)", n);
        return -1;
    }
    const size_t o = (size_t)off * ld + off;
    double *bA = w.A + o, *bW = w.W + o, *bX = w.Kinv + o;
    auto tile = [&](int i, int j) { return (size_t)(LB * i) * ld + (size_t)LB * j; };
    GPX_TRY(gpx_sweep_phase(s, w, off, T, aug, 0, !inverse));
    // Dense row panels (round 5, GPX_SWEEP_LITE=0: the round-4 phases): the tiles right of the
    // diagonal are solved by sweep_xs_kernel -- two workgroups a CU instead of one, each with
    // its tile in registers -- which also applies the last `depth` trailing updates of its tile
    // itself (all of them up to eight tiles: the rank-128 .. 896 products they replace moved
    // more bytes than they computed on, and every one was a launch on the group's chain);
    // earlier steps stay ONE product of the tile engine. The next diagonal tile takes its
    // updates in the same launch; what is left for sweep_kernel is its last update and the
    // leaf: two launches a tile row instead of four.
    const bool lite = gpx_sweep_lite();
    const int depth = gpx_sweep_fold_depth(T);
    const int ig = w.full_w && off == 0 && n == w.np && T > 8 ? T : 8;
    auto inverse_column = [&](hipStream_t st, int q) -> int {
        const int i0 = q / ig * ig;                        // inside the 1024-block of tile q (or all)
        if (q == i0) return 0;
        const int rows = LB * (q - i0);
        // T = W[i0:q, i0:q] R[i0:q, q] (W upper: k >= row tile), into the scratch
        GPX_TRY(gpx_gemm(st, 0, 0,
                         mk(bW + tile(i0, i0), ld, bA + tile(i0, q), ld, bX + tile(i0, q), ld, rows,
                            LB, rows, 1.0, 0.0, GEMM_KLO_M)));
        // W[i0:q, q] = -T W_qq (W_qq upper: k <= column)
        return gpx_gemm(st, 0, 0,
                        mk(bX + tile(i0, q), ld, bW + tile(q, q), ld, bW + tile(i0, q), ld, rows,
                           LB, LB, -1.0, 0.0, GEMM_KHI_N));
    };
    for (int q = 0; q < T; ++q) {
        if (lite) {
            // first tile column the dense launch solves: (q, q+1) too (GPX_SWEEP_PRE=0: that one
            // stays with the fused task, which then solves it as well)
            static const int pre = env_int("GPX_SWEEP_PRE", 1);
            const int t0 = q + 1 < T && !pre ? q + 2 : q + 1;
            // Small matrices (up to GPX_SWEEP_RIGHT tiles): RIGHT-looking -- the dense launch of
            // row q applies step q-1 to every tile from row q down (the same order per tile),
            // so that no launch carries a chain of more than one update in front of its solves:
            // the last rows are a handful of tiles with q updates each otherwise, on a device
            // that is three quarters idle
            if (pre && T <= gpx_sweep_right_max()) {
                GPX_TRY(gpx_sweep_xs(s, w, off, T, aug, q, t0, std::max(0, q - 1), 3));
                GPX_TRY(gpx_sweep_phase(s, w, off, T, aug, 1 + q, !inverse, true, true));
                continue;
            }
            const int kf = std::max(0, q - depth);
            // (the right-hand-side tile stays out of the products: narrow, it takes ALL its
            // steps in the dense launch at an eighth of the cost -- as one of TW - q - 1 full
            // tile columns of these products it was a quarter of their work at 16 tiles)
            static const int rhs_env = env_int("GPX_SWEEP_RHS_DENSE", 1);
            const bool rhs_dense = aug && rhs_env && gpx_sweep_narrow();
            const int ncols = TW - q - 1 - (rhs_dense ? 1 : 0);
            if (kf > 0) {                                  // the steps before kf: the tile engine
                if (ncols > 0)
                    GPX_TRY(gpx_gemm(s, 1, 0,
                                     mk(bA + tile(0, q), ld, bA + tile(0, q + 1), ld,
                                        bX + tile(q, q + 1), ld, LB, LB * ncols, LB * kf, -1.0,
                                        1.0, 0)));
                if (q + 1 < T)
                    GPX_TRY(gpx_gemm(s, 1, 0,
                                     mk(bA + tile(0, q + 1), ld, bA + tile(0, q + 1), ld,
                                        bA + tile(q + 1, q + 1), ld, LB, LB, LB * kf, -1.0, 1.0, 0)));
            }
            // steps kf .. q-1 of every tile right of the diagonal and of the next diagonal tile,
            // and the solves of the tiles from t0 on
            GPX_TRY(gpx_sweep_xs(s, w, off, T, aug, q, t0, kf, pre ? 1 : 2, rhs_dense ? 0 : -1));
            GPX_TRY(gpx_sweep_phase(s, w, off, T, aug, 1 + q, !inverse, true, pre != 0));
            continue;
        }
        if (q >= 1) {
            if (TW - q - 1 > 0)
                GPX_TRY(gpx_gemm(s, 1, 0,
                                 mk(bA + tile(0, q), ld, bA + tile(0, q + 1), ld,
                                    bX + tile(q, q + 1), ld, LB, LB * (TW - q - 1), LB * q, -1.0,
                                    1.0, 0)));
            if (q + 1 < T)
                GPX_TRY(gpx_gemm(s, 1, 0,
                                 mk(bA + tile(0, q + 1), ld, bA + tile(0, q + 1), ld,
                                    bA + tile(q + 1, q + 1), ld, LB, LB, LB * q, -1.0, 1.0, 0)));
        }
        if (TW - q - 1 > 0) GPX_TRY(gpx_sweep_phase(s, w, off, T, aug, 1 + q, !inverse));
    }
    if (!inverse) return 0;
    // Columns of R^-1 in BLOCKS of nb (round 5, GPX_SWEEP_INVBLOCK, 1: column by column as in
    // round 4): for the columns [b, b + nb) the part of the sums that needs only the columns
    // before the block -- T[i0:b, b:b+nb] = W[i0:b, i0:b] R[i0:b, b:b+nb], most of the flops --
    // is ONE product nb tiles wide; each column then continues ITS sum over the rows of the
    // block (beta = 1: the accumulator starts from the stored partial sum, k still ascending --
    // the chain of the one-product form, the same bits) and takes its product with W_qq.
    static const int nbenv = env_int("GPX_SWEEP_INVBLOCK", 4);
    const int nb = std::max(1, std::min(nbenv, 8));
    for (int b = 1; b < T;) {
        const int i0 = b / ig * ig;
        if (b == i0) {                                     // first tile of an inverse group: the leaf's
            ++b;
            continue;
        }
        const int e = std::min(std::min(b + nb, T), i0 + ig);   // the block [b, e), inside the group
        if (nb == 1 || e - b < 2) {
            GPX_TRY(inverse_column(s, b));
            ++b;
            continue;
        }
        const int rows0 = LB * (b - i0);
        // the wide product: rows i0 .. b-1, sums over k < b (W upper: k >= row tile)
        GPX_TRY(gpx_gemm(s, 0, 0,
                         mk(bW + tile(i0, i0), ld, bA + tile(i0, b), ld, bX + tile(i0, b), ld, rows0,
                            LB * (e - b), rows0, 1.0, 0.0, GEMM_KLO_M)));
        for (int q = b; q < e; ++q) {
            if (q > b) {
                const int kk = LB * (q - b);
                // rows above the block: the sum continues over the block's rows b .. q-1
                GPX_TRY(gpx_gemm(s, 0, 0,
                                 mk(bW + tile(i0, b), ld, bA + tile(b, q), ld, bX + tile(i0, q), ld,
                                    rows0, LB, kk, 1.0, 1.0, 0)));
                // rows of the block: their sums start there (W upper: k >= row tile)
                GPX_TRY(gpx_gemm(s, 0, 0,
                                 mk(bW + tile(b, b), ld, bA + tile(b, q), ld, bX + tile(b, q), ld, kk,
                                    LB, kk, 1.0, 0.0, GEMM_KLO_M)));
            }
            // W[i0:q, q] = -T W_qq (W_qq upper: k <= column)
            GPX_TRY(gpx_gemm(s, 0, 0,
                             mk(bX + tile(i0, q), ld, bW + tile(q, q), ld, bW + tile(i0, q), ld,
                                LB * (q - i0), LB, LB, -1.0, 0.0, GEMM_KHI_N)));
        }
        b = e;
    }
    return 0;
}

// R and (if inverse) W = R^-1 of the diagonal block (off, n)
static int potrf_rec(hipStream_t s, const DenseWs &w, int off, int n, bool inverse)
{
    const int ld = w.ld;
    const size_t o11 = (size_t)off * ld + off;
    if (n == LB) {
        return gpx_potrf_leaf2(s, w.A + o11, ld, w.W + o11, ld, w.info, off, w.batch, w.mstride);
    }
    // small blocks: factor and full inverse as one task-queue launch (panel.hip), or, for a
    // workspace of many members, phase by phase over all of them
    if (n <= gpx_panel_max(w.np) && w.pctl) {
        if (sweep_on(w)) return sweep_block(s, w, off, n, false, true);
        return gpx_panel(s, w, off, n);
    }
    const int n1 = split(n), n2 = n - n1;
    const size_t o12 = o11 + n1, o22 = (size_t)(off + n1) * ld + off + n1;
    if (wide_ok(w, n1, n2, 2)) {
        // the left half, R12 and the update of the right half in one wide panel launch;
        // every tile it touches is up to date when this block starts: no gates
        GPX_TRY(gpx_panel(s, w, off, n1, n2, w.gate_total[0], w.gate_total[1]));
    } else {
        // the left half always gets its full inverse: the panel step multiplies by it
        GPX_TRY(potrf_rec(s, w, off, n1, true));
        // R12 = W11^T A12, out of place: the off-diagonal tiles of this block have
        // lived in Kinv since they were built / last updated, R12 lands in A.
        // op(A)[m][k] = W11[k][m] is lower triangular: k < m0 + TILE
        {
            GemmArgs g = mk(w.W + o11, ld, w.Kinv + o12, ld, w.A + o12, ld, n1, n2, n1, 1.0,
                            0.0, GEMM_KHI_M);
            g.order = env_int("GPX_ORD_R12", 1);
            GPX_TRY(gpx_gemm(s, 1, 0, g));
        }
        // A22 -= R12^T R12, upper tiles only: diagonal tiles in A, the others in Kinv
        GPX_TRY(syrk_upper(s, w.A + o12, ld, w.A + o22, ld, n2, n1, w.Kinv + o22));
    }
    GPX_TRY(potrf_rec(s, w, off + n1, n2, inverse));
    if (inverse) GPX_TRY(extend_inverse(s, w, off, n, n1));
    return 0;
}

// ---- diagonal blocks of the right-looking driver ---------------------------------
int gpx_block_layout(int np, int *offs, bool full_inverse)
{
    static const int split_last = env_int("GPX_SPLIT_LAST", 0);
    static const int env0 = env_int("GPX_NB0", 0), env = env_int("GPX_NB", 0);
    static std::vector<int> list;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *e = getenv("GPX_BLOCKS");
        while (e && *e) {
            const int v = atoi(e);
            if (v >= LB && v % LB == 0) list.push_back(v);
            e = strchr(e, ',');
            if (e) ++e;
        }
    });
    int count = 0, at = 0;
    offs[0] = 0;
    if (!list.empty()) {
        for (size_t i = 0; at < np; ++i) {
            int b = list[i < list.size() ? i : list.size() - 1];
            if (count == GPX_MAX_BLOCKS - 1) b = np - at;
            at = std::min(np, at + b);
            offs[++count] = at;
        }
        return count;
    }
    const int a = env0 >= LB && env0 % LB == 0 ? env0 : 1024;
    int b = env >= LB && env % LB == 0 ? env : (np > 8192 ? 2048 : 1024);
    while (1 + (np - a + b - 1) / b > GPX_MAX_BLOCKS) b *= 2;
    at = std::min(np, a);
    offs[++count] = at;
    while (at < np) {
        at = std::min(np, at + b);
        offs[++count] = at;
    }
    if (full_inverse && split_last && np >= 4096 && np < 8192 && count < GPX_MAX_BLOCKS &&
        offs[count] - offs[count - 1] == 1024) {
        offs[count + 1] = offs[count];
        offs[count] = offs[count - 1] + 512;
        ++count;
    }
    return count;
}
typedef GpxBlocks Blocks;

// W = R^-1 over the blocks [b0, b1): both halves, then the (1,2) block of the node
static int trtri_blocks(hipStream_t s, const DenseWs &w, const Blocks &bl, int b0, int b1)
{
    if (b1 - b0 < 2) return 0;
    const int mid = (b0 + b1) / 2;
    GPX_TRY(trtri_blocks(s, w, bl, b0, mid));
    GPX_TRY(trtri_blocks(s, w, bl, mid, b1));
    return extend_inverse(s, w, bl.off(b0), bl.off(b1) - bl.off(b0), bl.off(mid) - bl.off(b0));
}

#define GPX_EV(expr) GPX_HIP(expr)

// Block column k of the inverse and its contribution to (R^T R)^-1:
//   P       = W[:k, :k] R[:k, k]                  (W upper: j >= row tile; scratch: Kinv[:k, k])
//   W[:k,k] = -P W_kk                             (W_kk upper: j <= column tile)
//   Kinv[:k+1, :k+1] (+)= Wc Wc^T, Wc = W[:k+1, k]  (upper tiles; the new block column
//                                                  starts from zero)
// Summed over k these are the N^3/3 + N^3/3 flops of trtri + lauum as rank-NB products
// that only trail the factorisation by one block, so they fill the GPU while the last
// diagonal blocks (a latency-bound chain with little trailing matrix left) are factored.
// Round 3 reassociated the column: rounds 1-2 formed T = R[:k, k] W_kk first and then
// -W[:k, :k] T, so the large product (k^2 nk flops) waited for the diagonal block's
// inverse. P needs only rows < k of R (final when row panel k-1 is: `early`) and runs
// BESIDE the factorisation of block k; what waits for W_kk (`late`) is the small product.
// For the last block that takes the large product off the end of the evaluation, where
// nothing hides it. Same flops, same launch shapes; which association a matrix gets
// depends on its size only. GPX_INVCOL_EARLY=0 / 1 forces one (last bits differ).
static int inverse_column(hipStream_t s, const DenseWs &w, const Blocks &bl, int k, bool kinv,
                          hipEvent_t after_w = nullptr, hipEvent_t early = nullptr,
                          hipEvent_t late = nullptr)
{
    // default: up to np = 8192, where the chain of diagonal blocks is the critical path
    // (N = 2048 1.13 -> 1.09 ms, 4096 2.64 -> 2.55, 8192 10.90 -> 10.87); above, the large
    // product beside the diagonal block only takes CUs from it (N = 16384 69.3 -> 69.8 ms)
    static const int early_env = env_int("GPX_INVCOL_EARLY", -1);
    const bool early_on = early_env >= 0 ? early_env != 0 : w.np <= 8192;
    const int ld = w.ld, ok = bl.off(k), nk = bl.len(k);
    const size_t okk = (size_t)ok * ld + ok;
    if (k > 0 && early_on) {
        if (early) GPX_EV(hipStreamWaitEvent(s, early, 0));
        GPX_TRY(gpx_gemm(s, 0, 0,
                         mk(w.W, ld, w.A + ok, ld, w.Kinv + ok, ld, ok, nk, ok, 1.0, 0.0,
                            GEMM_KLO_M | (env_int("GPX_KREV_INVCOL", 0) ? GEMM_KREV : 0))));
        if (late) GPX_EV(hipStreamWaitEvent(s, late, 0));
        GemmArgs g = mk(w.Kinv + ok, ld, w.W + okk, ld, w.W + ok, ld, ok, nk, nk, -1.0, 0.0,
                        GEMM_KHI_N);
        g.order = env_int("GPX_ORD_T", 2);
        GPX_TRY(gpx_gemm(s, 0, 0, g));
    } else {
        if (late) GPX_EV(hipStreamWaitEvent(s, late, 0));
        if (k > 0) {
            {
                GemmArgs g = mk(w.A + ok, ld, w.W + okk, ld, w.Kinv + ok, ld, ok, nk, nk, 1.0, 0.0,
                                GEMM_KHI_N);
                g.order = env_int("GPX_ORD_T", 2);
                GPX_TRY(gpx_gemm(s, 0, 0, g));
            }
            GPX_TRY(gpx_gemm(s, 0, 0,
                             mk(w.W, ld, w.Kinv + ok, ld, w.W + ok, ld, ok, nk, ok, -1.0, 0.0,
                                GEMM_KLO_M | (env_int("GPX_KREV_INVCOL", 0) ? GEMM_KREV : 0))));
        }
    }
    if (after_w) GPX_HIP(hipEventRecord(after_w, s));        // block column k of R^-1 is in
    if (!kinv) return 0;
    // Kinv[i][j] += sum_c Wc[i][c] Wc[j][c]; inside the diagonal block W_kk is upper
    // triangular: Wc[m][c] == 0 for c < m - ok
    const int n = ok + nk;
    GemmArgs g = mk(w.W + ok, ld, w.W + ok, ld, w.Kinv, ld, n, n, nk, 1.0, 1.0,
                    GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KLO_N);
    g.kshift = ok;
    g.beta0_from = ok;
    return gpx_gemm(s, 0, 1, g);
}

// Value-only factorisation of a small matrix (round 3): every tile of it as a task of
// ONE panel launch -- the chain of diagonal tiles runs through without the two
// dependent product launches between 1024-blocks (0.17 ms each beside 0.35-ms
// blocks), the trailing updates of all steps are tasks of the same launch. What it
// leaves behind is what the blocked sweep leaves: R, and the inverses of the 1024-
// blocks of the default partition (the panel assembles W inside those only), so that
// everything that follows -- substitution by blocks, the completion of the inverse --
// finds the same state. Only when nothing else runs on the device: the launch takes
// up to 253 CUs. GPX_PANEL_WHOLE=<largest order> (0: off).
bool gpx_potrf_whole(const DenseWs &w, int mode)
{
    static const int whole_max = [] {
        const int v = env_int("GPX_PANEL_WHOLE", GPX_PANEL_WHOLE_DEFAULT);
        return v < 0 ? 0 : (v > GPX_PANEL_WHOLE_MAX ? GPX_PANEL_WHOLE_MAX : v);
    }();
    if (mode != GPX_POTRF_R || w.np > whole_max || w.np <= GPX_PANEL_MAX || !w.pctl) return false;
    const Blocks bl(w.np, false);
    bool regular = true;                               // blocks of 1024, the last one any
    for (int k = 0; k < bl.count; ++k)
        regular = regular && bl.off(k) == 1024 * k && (k == bl.count - 1 || bl.len(k) == 1024);
    // (round 3 also asked that nothing else ran on the device; since round 4 the choice
    // depends on the matrix alone, so that a member of a batch and the same evaluation on
    // its own take the same order of arithmetic)
    return regular && gpx_panel_max(w.np) >= 1024 && gpx_panel_streaming();
}

// The factorisation mode of an evaluation WITH gradients (round 5). Up to np = 4096 the ONE
// whole-matrix launch (value-only mode; the chain of diagonal tiles runs through, 31 us per
// tile since round 5) followed by trtri + lauum on the tile engine beats the sweep over
// 1024-blocks with R^-1 and K^-1 inside it (four panel launches with two dependent product
// launches between each pair): N = 1536 0.83 -> 0.72 ms, 2048 1.03 -> 0.91, 3072 / 4096 level.
// A function of the padded order alone -- single evaluations and members of groups ask here
// -- so that lZ has the bits of the value-only evaluation up to np = 4096 and a member keeps
// the bits of its single evaluation. GPX_GRAD_WHOLE = largest padded order (0: never).
int gpx_grad_mode(const DenseWs &w)
{
    static const int grad_whole = env_int("GPX_GRAD_WHOLE", 4096);
    if (w.np <= grad_whole && gpx_potrf_whole(w, GPX_POTRF_R)) return GPX_POTRF_R;
    return GPX_POTRF_KINV;
}

// ... and on that route (up to np = GPX_GRAD_FULL_W, 4096) the factorisation -- one launch, or one
// lock-step sweep, over the whole matrix -- assembles ALL of R^-1 beside R instead of leaving
// the completion to gpx_trtri: the columns of the inverse as chunked sums on the launch's
// workers (panel.hip, Graph::build). One evaluation with gradients, off -> on: N = 1280
// 0.604 -> 0.549 ms, 2048 0.893 -> 0.83, 2560 1.228 -> 1.127, 3072 1.51 -> 1.45, 3584
// 1.94 -> 1.80, 4096 2.39 -> 2.285. Groups of 8 members in ONE launch share its workers and
// lose: -1 % at 1536, -9 % at 2048, -10 % at 3072, -13 % at 4096; lock-step sweeps (the sums
// as one product on the tile engine) within 3 % either way. A function of np alone, as
// gpx_grad_mode, so that members keep the bits of single evaluations -- the reference's
// primary pattern (optimize()). profiles/r05_full_w_ab.txt.
bool gpx_grad_full_w(const DenseWs &w, int mode)
{
    static const int full_max = env_int("GPX_GRAD_FULL_W", 4096);
    return w.np <= full_max && mode == GPX_POTRF_R && Blocks(w.np).count > 1 &&
           gpx_potrf_whole(w, mode);
}

bool gpx_potrf_rhs_ok(const DenseWs &w, int mode)
{
    static const bool on = !(getenv("GPX_PANEL_RHS") && !atoi(getenv("GPX_PANEL_RHS")));
    if (!on || w.ld < w.np + LB || !w.pctl) return false;
    if (gpx_potrf_whole(w, mode)) return true;
    // one block that is one panel launch (or one sweep) of at least two tiles
    return Blocks(w.np, mode == GPX_POTRF_KINV).count == 1 && w.np >= 2 * LB &&
           w.np <= gpx_panel_max(w.np) && gpx_panel_streaming();
}

int gpx_potrf(hipStream_t s, const DenseWs &w, int mode, bool offdiag_staged)
{
    if (w.np % LB || w.ld < w.np || w.ld % 2 || !w.A || !w.W || !w.Kinv) {
        gpx_set_error("potrf: bad workspace (order %d)", w.np);
        return -1;
    }
    if (!offdiag_staged) {
        if (w.batch > 1) {
            gpx_set_error("potrf: member-batched workspaces come with their tiles staged");
            return -1;
        }
        GPX_TRY(copy_block(s, w.A, w.Kinv, w.ld, w.np, w.np));
    }
    const BatchScope batch_scope(w);
    const Blocks bl(w.np, mode == GPX_POTRF_KINV);
    const int nb = bl.count, ld = w.ld;
    if (nb == 1) {                                     // one block: its inverse is W
        if (w.aug_rhs) {
            // ... and one panel, with the caller's right-hand side as one more tile column
            if (!gpx_potrf_rhs_ok(w, mode)) {
                gpx_set_error("potrf: no room for a right-hand side beside this matrix");
                return -1;
            }
            if (sweep_on(w)) GPX_TRY(sweep_block(s, w, 0, w.np, true, !w.no_inverse));
            else GPX_TRY(gpx_panel(s, w, 0, w.np, LB, 0, 0, true));
        } else {
            GPX_TRY(potrf_rec(s, w, 0, w.np, true));
        }
        return mode == GPX_POTRF_KINV ? gpx_lauum(s, w) : 0;
    }
    // one launch over the whole matrix: the caller decided (gpx_potrf_whole) and, with
    // aug_rhs, has put the right-hand side in place -- never decided again here
    if (w.whole) {
        if (!gpx_potrf_whole(w, mode)) {
            gpx_set_error("potrf: a whole-matrix launch was asked for a matrix it cannot take");
            return -1;
        }
        if (sweep_on(w)) return sweep_block(s, w, 0, w.np, w.aug_rhs, !w.no_inverse);
        return gpx_panel(s, w, 0, w.np, w.aug_rhs ? 128 : 0);
    }
    if (w.aug_rhs) {
        gpx_set_error("potrf: a right-hand side rides along with whole-matrix launches only");
        return -1;
    }
    // look-ahead needs the extra streams and two events per block
    const bool ahead = w.crit && w.bulk && w.aux && w.events && nb <= GPX_MAX_BLOCKS &&
                       w.batch <= 1;
    // Only with a CU partition (GPX_RESERVE_CUS > 0; w.crit_only is null otherwise):
    // with the inverse in the same sweep every diagonal block above np = 8192 hides
    // completely under the products of its step and gets the reserved CUs and nothing
    // else (strict partition, 1.7 ms faster at N = 16384 than "anywhere, first in
    // line"). Otherwise the diagonal blocks may run anywhere.
    const bool strict = ahead && w.crit_only && mode != GPX_POTRF_R && w.np > 8192;
    static const int overlap_env = env_int("GPX_OVERLAP_NOSPLIT", 1);
    struct OverlapScope {
        explicit OverlapScope(int v) { tl_overlap = v; }
        ~OverlapScope() { tl_overlap = 0; }
    } overlap_scope(ahead && overlap_env ? 1 : 0);
    hipStream_t crit = ahead ? (strict ? w.crit_only : w.crit) : s;
    hipStream_t bulk = ahead ? w.bulk : s;
    // The inverse columns run on a third stream (same CUs as `bulk`, low priority) beside the
    // trailing updates of their step: the two launch sequences fill each other's
    // drain tails (75.9 against 76.8 ms per evaluation at N = 16384; with 248 instead
    // of 224 CUs for the products it was the other way round, the third stream's
    // backlog ending up serial). GPX_AUX=0: they follow the trailing update on `bulk`.
    static const int aux_on = env_int("GPX_AUX", 1);
    hipStream_t aux = ahead && aux_on ? w.aux : bulk;
    const int slots = ahead ? w.bulk_slots : 0;
    hipEvent_t *F = w.events, *D = w.events + GPX_MAX_BLOCKS;
    hipEvent_t evJoin = ahead ? w.events[4 * GPX_MAX_BLOCKS] : nullptr;
    hipEvent_t evAux = ahead ? w.events[4 * GPX_MAX_BLOCKS + 1] : nullptr;
    hipEvent_t evW = ahead ? w.events[4 * GPX_MAX_BLOCKS + 2] : nullptr;
    // the caller goes on (vector kernels that need R and R^-1 only) while the last K^-1
    // update is still running
    const bool defer = ahead && w.defer_kinv && mode == GPX_POTRF_KINV && aux != bulk;
    // Without the strict partition the diagonal blocks are (part of) the critical path,
    // and on `bulk` the two small products between F_k and F_k+1 -- the block column of
    // the row panel that the next diagonal block needs and that block's update -- queued
    // behind the far trailing update of step k-1 (N = 8192: 0.5-0.8 ms between two 0.45-ms
    // diagonal blocks). Fast chain: they run on `crit` itself, right behind F_k (high
    // priority, every CU), the far update of a step hands over its first diagonal block
    // (event TD) before it goes on, and `bulk` keeps the rest.
    static const int fast_env = env_int("GPX_FASTCHAIN", 1);
    const bool fast = ahead && !strict && fast_env != 0;
    hipEvent_t *G = w.events + 2 * GPX_MAX_BLOCKS;       // row k+1 carries update k
    hipEvent_t *TD = w.events + 3 * GPX_MAX_BLOCKS;      // block (k+2,k+2) carries update k
    bool lead = false;
    if (ahead) {
        GPX_EV(hipEventRecord(D[0], s));               // the build of the matrix is in
        // the first diagonal block only needs its own rows (a wide panel: also the next
        // block's)
        const bool wide0 = nb > 1 && wide_ok(w, bl.len(0), bl.len(1), 1);
        lead = w.lead && w.lead_rows >= (wide0 ? bl.off(2) : bl.len(0));
        GPX_EV(hipStreamWaitEvent(crit, lead ? w.lead : D[0], 0));
        GPX_EV(hipStreamWaitEvent(bulk, D[0], 0));
        if (aux != bulk) GPX_EV(hipStreamWaitEvent(aux, D[0], 0));
    }
    for (int k = 0; k < nb; ++k) {
        const int ok = bl.off(k), nk = bl.len(k);
        // F_k: R_kk and W_kk = R_kk^-1 (what the row panel multiplies by)
        if (ahead && k > 0) GPX_EV(hipStreamWaitEvent(crit, D[k], 0));
        // wide: with R[k, k+1] and update k of block (k+1, k+1) in the same launch; its
        // tasks on those tiles wait for the gates as they stand now (every launch that
        // moves them for this step has been enqueued)
        const bool wide = k + 1 < nb && wide_ok(w, nk, bl.len(k + 1), 1);
        if (wide)
            GPX_TRY(gpx_panel(crit, w, ok, nk, bl.len(k + 1), w.gate_total[0], w.gate_total[1]));
        else
            GPX_TRY(potrf_rec(crit, w, ok, nk, true));
        if (ahead) {
            GPX_EV(hipEventRecord(F[k], crit));
            GPX_EV(hipStreamWaitEvent(bulk, F[k], 0));    // (aux: inside inverse_column)
            // F_0 ran on the rows of its own build; whatever `crit` does next (the update
            // of the next diagonal block, which the lead rows need not cover) waits for
            // the whole matrix. Found in round 3 as a race: with three batch members
            // competing the rest of the build can still be running when F_0 is done, and
            // block (1,1) was updated before it was written (one evaluation at a time the
            // build finishes long before F_0 does).
            if (lead && k == 0) GPX_EV(hipStreamWaitEvent(crit, D[0], 0));
        }
        if (fast && k + 1 < nb) {
            const int o1 = bl.off(k + 1), n1 = bl.len(k + 1), rest = w.np - o1;
            const size_t okk = (size_t)ok * ld + ok, ok1 = (size_t)ok * ld + o1;
            const size_t o11 = (size_t)o1 * ld + o1;
            // crit: R[k,k+1] = W_kk^T X[k,k+1], then update k of block (k+1,k+1) (a wide
            // panel has done both)
            if (!wide) {
                if (k >= 1) GPX_EV(hipStreamWaitEvent(crit, G[k - 1], 0));
                {
                    GemmArgs g = mk(w.W + okk, ld, w.Kinv + ok1, ld, w.A + ok1, ld, nk, n1, nk,
                                    1.0, 0.0, GEMM_KHI_M);
                    g.order = env_int("GPX_ORD_R12", 1);
                    GPX_TRY(gpx_gemm(crit, 1, 0, g));
                }
                if (k >= 1) GPX_EV(hipStreamWaitEvent(crit, TD[k - 1], 0));
                GPX_TRY(syrk_upper(crit, w.A + ok1, ld, w.A + o11, ld, n1, nk, w.Kinv + o11));
            }
            GPX_EV(hipEventRecord(D[k + 1], crit));
            // bulk: the rest of the row panel ...
            if (rest > n1) {
                GemmArgs g = mk(w.W + okk, ld, w.Kinv + ok1 + n1, ld, w.A + ok1 + n1, ld, nk,
                                rest - n1, nk, 1.0, 0.0, GEMM_KHI_M);
                g.order = env_int("GPX_ORD_R12", 1);
                g.slots = slots;
                GPX_TRY(gpx_gemm(bulk, 1, 0, g));
            }
            GPX_EV(hipStreamWaitEvent(bulk, D[k + 1], 0));       // R[k,k+1] is in A
            if (k + 2 < nb) {
                const int o2 = bl.off(k + 2), n2 = bl.len(k + 2), rest2 = w.np - o2;
                const size_t ok2 = (size_t)ok * ld + o2, o12 = (size_t)o1 * ld + o2,
                             o22 = (size_t)o2 * ld + o2;
                // ... the off-diagonal tiles of row k+1 ...
                {
                    GemmArgs g = mk(w.A + ok1, ld, w.A + ok2, ld, w.Kinv + o12, ld, n1, rest2,
                                    nk, -1.0, 1.0, 0);
                    g.slots = slots;
                    GPX_TRY(gpx_gemm(bulk, 1, 0, g));
                }
                GPX_TRY(gate_bump(bulk, w, 0));                  // row k+1 carries update k
                GPX_EV(hipEventRecord(G[k], bulk));
                // ... and the trailing blocks, the diagonal block of step k+2 first
                GPX_TRY(syrk_upper(bulk, w.A + ok2, ld, w.A + o22, ld, n2, nk, w.Kinv + o22,
                                   slots));
                GPX_TRY(gate_bump(bulk, w, 1));                  // block (k+2, k+2) too
                GPX_EV(hipEventRecord(TD[k], bulk));
                if (k + 3 < nb) {
                    const int o3 = bl.off(k + 3), rest3 = w.np - o3;
                    const size_t ok3 = (size_t)ok * ld + o3, o23 = (size_t)o2 * ld + o3,
                                 o33 = (size_t)o3 * ld + o3;
                    GemmArgs g = mk(w.A + ok2, ld, w.A + ok3, ld, w.Kinv + o23, ld, n2, rest3, nk,
                                    -1.0, 1.0, 0);
                    g.slots = slots;
                    GPX_TRY(gpx_gemm(bulk, 1, 0, g));
                    GPX_TRY(syrk_upper(bulk, w.A + ok3, ld, w.A + o33, ld, rest3, nk,
                                       w.Kinv + o33, slots));
                }
            }
        } else if (k + 1 < nb) {
            const int o1 = bl.off(k + 1), n1 = bl.len(k + 1), rest = w.np - o1;
            const size_t okk = (size_t)ok * ld + ok, ok1 = (size_t)ok * ld + o1;
            // row panel R[k, k+1:] = W_kk^T A[k, k+1:], out of place: the off-diagonal
            // tiles of row k have lived in Kinv since they were built / last updated, R
            // lands in A. op(A)[m][j] = W_kk[j][m] is lower triangular: j < m0 + TILE
            // (a wide panel has the first n1 columns and the next diagonal block's update)
            const int skip = wide ? n1 : 0;
            if (rest > skip) {
                GemmArgs g = mk(w.W + okk, ld, w.Kinv + ok1 + skip, ld, w.A + ok1 + skip, ld, nk,
                                rest - skip, nk, 1.0, 0.0, GEMM_KHI_M);
                g.order = env_int("GPX_ORD_R12", 1);
                g.slots = slots;
                GPX_TRY(gpx_gemm(bulk, 1, 0, g));
            }
            // update k of the next diagonal block first: F_k+1 can start
            const size_t o11 = (size_t)o1 * ld + o1;
            if (!wide)
                GPX_TRY(syrk_upper(bulk, w.A + ok1, ld, w.A + o11, ld, n1, nk, w.Kinv + o11,
                                   slots));
            if (ahead) GPX_EV(hipEventRecord(D[k + 1], bulk));
            if (k + 2 < nb) {
                const int o2 = bl.off(k + 2), rest2 = w.np - o2;
                const size_t ok2 = (size_t)ok * ld + o2, o12 = (size_t)o1 * ld + o2,
                             o22 = (size_t)o2 * ld + o2;
                // ... then the off-diagonal tiles of row k+1 (they stay in the staging
                // area) ...
                {
                    GemmArgs g = mk(w.A + ok1, ld, w.A + ok2, ld, w.Kinv + o12, ld, n1, rest2,
                                    nk, -1.0, 1.0, 0);
                    g.slots = slots;
                    GPX_TRY(gpx_gemm(bulk, 1, 0, g));
                }
                GPX_TRY(gate_bump(bulk, w, 0));
                // ... and the trailing blocks: diagonal tiles in A, the others in Kinv
                GPX_TRY(syrk_upper(bulk, w.A + ok2, ld, w.A + o22, ld, rest2, nk, w.Kinv + o22,
                                   slots));
                GPX_TRY(gate_bump(bulk, w, 1));
            }
        }
        // the inverse follows one block behind, on its own stream
        if (mode != GPX_POTRF_R)
            GPX_TRY(inverse_column(aux, w, bl, k, mode == GPX_POTRF_KINV,
                                   defer && k == nb - 1 ? evW : nullptr,
                                   ahead && aux != bulk && k > 0 ? D[k] : nullptr,
                                   ahead && aux != bulk ? F[k] : nullptr));
    }
    if (ahead) {                                       // back on the caller's stream
        GPX_EV(hipEventRecord(evJoin, bulk));
        GPX_EV(hipStreamWaitEvent(s, evJoin, 0));
        if (aux != bulk) {
            GPX_EV(hipEventRecord(evAux, aux));
            GPX_EV(hipStreamWaitEvent(s, defer ? evW : evAux, 0));
        }
    }
    return 0;
}

int gpx_potrf_join(hipStream_t s, const DenseWs &w)
{
    // evAux was recorded by the last gpx_potrf of this workspace (waiting for an event
    // that has completed, or was never recorded, costs nothing)
    if (w.defer_kinv && w.events && w.aux)
        GPX_EV(hipStreamWaitEvent(s, w.events[4 * GPX_MAX_BLOCKS + 1], 0));
    return 0;
}

hipEvent_t gpx_potrf_lead_event(const DenseWs &w)
{
    return w.events ? w.events[4 * GPX_MAX_BLOCKS + 3] : nullptr;
}

// after potrf(..., false): W holds the inverses of the diagonal blocks only
int gpx_trtri(hipStream_t s, const DenseWs &w)
{
    const BatchScope batch_scope(w);
    const Blocks bl(w.np);
    return trtri_blocks(s, w, bl, 0, bl.count);
}

// X = R^-T B for B (np x m at Bp, ld ldb) in place, using the inverses of the
// diagonal blocks that a value-only potrf leaves behind: block forward substitution
// X_k = W_kk^T B_k through the scratch T (np x m, ld ldb), B[k+1:] -= R[k, k+1:]^T X_k.
int gpx_trsm_rt(hipStream_t s, const DenseWs &w, double *B, double *T, int ldb, int m,
                long long pstride)
{
    const BatchScope batch_scope(w);
    const Blocks bl(w.np);
    const int ld = w.ld;
    const int nb = w.batch > 1 ? w.batch : 1;
    // operands out of the members' panels move by pstride, not by the matrix stride
    auto panel = [&](GemmArgs g, bool a_panel) {
        if (nb > 1) {
            if (a_panel) g.strideA = pstride;
            g.strideB = pstride;
            g.strideC = pstride;
        }
        return g;
    };
    for (int k = 0; k < bl.count; ++k) {
        const int ok = bl.off(k), nk = bl.len(k);
        const size_t okk = (size_t)ok * ld + ok;
        double *Bk = B + (size_t)ok * ldb, *Tk = T + (size_t)ok * ldb;
        if (nk == LB) {
            // single row tile: the in-place multiply is safe with 128-tiles (each
            // workgroup reads its whole K=128 column panel before it writes)
            GemmArgs g = mk(w.W + okk, ld, Bk, ldb, Bk, ldb, LB, m, LB, 1.0, 0.0, 0);
            g.tile = 128;
            GPX_TRY(gpx_gemm(s, 1, 0, panel(g, false)));
        } else {
            GPX_TRY(copy_block(s, Bk, Tk, ldb, nk, m, nb, pstride));
            GemmArgs g = mk(w.W + okk, ld, Tk, ldb, Bk, ldb, nk, m, nk, 1.0, 0.0, GEMM_KHI_M);
            g.order = 1;
            GPX_TRY(gpx_gemm(s, 1, 0, panel(g, false)));
        }
        const int o1 = bl.off(k + 1), rest = w.np - o1;
        if (rest > 0)
            GPX_TRY(gpx_gemm(s, 1, 0,
                             panel(mk(w.A + (size_t)ok * ld + o1, ld, Bk, ldb, B + (size_t)o1 * ldb,
                                      ldb, rest, m, nk, -1.0, 1.0, 0), false)));
    }
    return 0;
}

int gpx_lauum(hipStream_t s, const DenseWs &w)
{
    // Kinv[i][j] = sum_{k >= max(i,j)} W[i][k] W[j][k], tiles with j >= i
    const BatchScope batch_scope(w);
    const int n = w.np, ld = w.ld;
    GemmArgs g = mk(w.W, ld, w.W, ld, w.Kinv, ld, n, n, n, 1.0, 0.0,
                    GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KLO_N |
                        (env_int("GPX_KREV", 0) ? GEMM_KREV : 0));
    g.order = env_int("GPX_ORD_LAUUM", 0);
    return gpx_gemm(s, 0, 1, g);
}
