// Blocked Cholesky, triangular inverse and symmetric inverse on the fp64 tile
// engine (gemm_f64.hip). Everything works on the UPPER factor R of the
// reference (R^T R = K + sn2 I, scipy.linalg.cholesky at
// /root/reference/pygp/inference/exact.py:54) in row-major storage.
//
//   potrf  recursive: factor the leading half AND invert its factor, get the row
//          panel as one triangle-aware MFMA GEMM R12 = W11^T A12 (W11 = R11^-1;
//          no substitution anywhere), SYRK the trailing half (upper tiles only),
//          recurse, then extend the inverse: W12 = -W11 (R12 W22). The inverse
//          of a left half is what its parent's panel step multiplies by, and it
//          is a block of the final R^-1 that exact.py:129 needs anyway, so
//          potrf + trtri cost 2N^3/3 flops in ~5 launches per tree node.
//          The 128x128 diagonal leaves are factored AND inverted by one
//          workgroup (leaf.hip).
//   lauum  Kinv = W W^T, upper tiles only, one launch with per-tile k ranges.
//
// potrf + trtri + lauum = N^3 flops, against the 7N^3/3 of the reference's
// cho_solve(R, eye(N)) route (exact.py:129).

#include "gpx_internal.h"

#define LB GPX_TILE                    // leaf order

// ---- environment knobs (developer experiments) --------------------------------
#include <cstdlib>
#include <map>
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
                                                         int rows, int cols)
{
    const int c2 = blockIdx.x * 256 + threadIdx.x;          // double2 column index
    if (2 * c2 >= cols) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        reinterpret_cast<double2 *>(dst + (size_t)r * ld)[c2] =
            reinterpret_cast<const double2 *>(src + (size_t)r * ld)[c2];
}

static int copy_block(hipStream_t s, const double *src, double *dst, int ld, int rows,
                      int cols)
{
    dim3 grid((cols / 2 + 255) / 256, rows < 1024 ? rows : 1024);
    hipLaunchKernelGGL(copy_block_kernel, grid, dim3(256), 0, s, src, dst, ld, rows, cols);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- recursion ---------------------------------------------------------------
static inline int split(int n) { return (n / LB / 2) * LB; }   // leading half

static GemmArgs mk(const double *A, int lda, const double *B, int ldb, double *C,
                   int ldc, int M, int N, int K, double alpha, double beta, int flags)
{
    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.strideA = g.strideB = g.strideC = 0;
    g.batch = 1;
    g.flags = flags;
    g.tile = 0;
    g.order = 0;
    g.swizzle = env_int("GPX_SWIZZLE", 0);
    g.waves = 0;
    g.use_lists = env_int("GPX_TILE_LISTS", 1);
    g.tiles = nullptr;
    return g;
}

// C -= P^T P on the upper tiles of the n x n block C, P = k x n. All live tiles
// cost the same, so a launch runs in ceil(tiles / slots) rounds and a count just
// above a multiple of the slot count (2080 tiles on 512 slots at n = 8192)
// wastes most of a round. Split the tile rows: the top rows fill whole rounds
// with 128-tiles, the short remainder goes out as 64-tiles (4x the workgroups,
// a quarter of the time each).
static int syrk_upper(hipStream_t s, const double *P, int ldp, double *C, int ldc, int n,
                      int k, double *C2)
{
    const int T = n / LB;
    const int slots = 512;                       // 256 CUs x 2 workgroups
    const long long live = (long long)T * (T + 1) / 2;
    int rows_top = T;
    if (env_int("GPX_SYRK_SPLIT", 1) && live > slots && live % slots != 0) {
        const long long whole = live / slots * slots;
        long long acc = 0;
        rows_top = 0;
        while (rows_top < T && acc + (T - rows_top) <= whole) acc += T - rows_top++;
        if (T - rows_top > T / 2) rows_top = T;  // not worth it
    }
    if (rows_top > 0) {
        GemmArgs g = mk(P, ldp, P, ldp, C, ldc, rows_top * LB, n, k, -1.0, 1.0,
                        GEMM_UPPER_ONLY);
        g.C2 = C2;
        GPX_TRY(gpx_gemm(s, 1, 0, g));
    }
    if (rows_top < T) {
        const int o = rows_top * LB;
        GemmArgs g = mk(P + o, ldp, P + o, ldp, C + (size_t)o * ldc + o, ldc, n - o, n - o,
                        k, -1.0, 1.0, GEMM_UPPER_ONLY);
        g.C2 = C2 ? C2 + (size_t)o * ldc + o : nullptr;
        g.tile = 64;
        GPX_TRY(gpx_gemm(s, 1, 0, g));
    }
    return 0;
}

// W12 = -W11 (R12 W22) for the node (off, n); Kinv's (1,2) block is scratch
static int extend_inverse(hipStream_t s, const DenseWs &w, int off, int n)
{
    const int ld = w.ld;
    const int n1 = split(n), n2 = n - n1;
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

// R and (if inverse) W = R^-1 of the diagonal block (off, n)
static int potrf_rec(hipStream_t s, const DenseWs &w, int off, int n, bool inverse)
{
    const int ld = w.ld;
    const size_t o11 = (size_t)off * ld + off;
    if (n == LB) {
        return gpx_potrf_leaf2(s, w.A + o11, ld, w.W + o11, ld, w.info, off);
    }
    // small blocks: factor and full inverse as one task-queue launch (panel.hip)
    if (n <= gpx_panel_max(w.np) && w.pctl) return gpx_panel(s, w, off, n);
    const int n1 = split(n), n2 = n - n1;
    const size_t o12 = o11 + n1, o22 = (size_t)(off + n1) * ld + off + n1;
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
    GPX_TRY(potrf_rec(s, w, off + n1, n2, inverse));
    if (inverse) GPX_TRY(extend_inverse(s, w, off, n));
    return 0;
}

int gpx_potrf(hipStream_t s, const DenseWs &w, bool full_inverse, bool offdiag_staged)
{
    if (w.np % LB || w.ld < w.np || w.ld % 2 || !w.A || !w.W || !w.Kinv) {
        gpx_set_error("potrf: bad workspace (order %d)", w.np);
        return -1;
    }
    if (!offdiag_staged) GPX_TRY(copy_block(s, w.A, w.Kinv, w.ld, w.np, w.np));
    return potrf_rec(s, w, 0, w.np, full_inverse);
}

// after potrf(..., false): the right spine still lacks its (1,2) inverse blocks
static int trtri_rec(hipStream_t s, const DenseWs &w, int off, int n)
{
    if (n == LB) return 0;
    // panels invert their whole block
    if (n <= gpx_panel_max(w.np) && w.pctl) return 0;
    const int n1 = split(n);
    GPX_TRY(trtri_rec(s, w, off + n1, n - n1));
    return extend_inverse(s, w, off, n);
}

int gpx_trtri(hipStream_t s, const DenseWs &w)
{
    return trtri_rec(s, w, 0, w.np);
}

// X = R^-T B for B (np x m at Bp, ld ldb) in place, using the inverses that a
// value-only potrf leaves behind (every left half): X1 = W11^T B1 through the
// scratch T (np x m, ld ldb), B2 -= R12^T X1, recurse into the right half.
static int trsm_rt_rec(hipStream_t s, const DenseWs &w, int off, int n, double *Bp,
                       double *Tp, int ldb, int m)
{
    const int ld = w.ld;
    const size_t o11 = (size_t)off * ld + off;
    if (n == LB) {
        // single row tile: the in-place multiply is safe with 128-tiles (each
        // workgroup reads its whole K=128 column panel before it writes)
        GemmArgs g = mk(w.W + o11, ld, Bp, ldb, Bp, ldb, LB, m, LB, 1.0, 0.0, 0);
        g.tile = 128;
        return gpx_gemm(s, 1, 0, g);
    }
    const int n1 = split(n), n2 = n - n1;
    GPX_TRY(copy_block(s, Bp, Tp, ldb, n1, m));
    {
        GemmArgs g = mk(w.W + o11, ld, Tp, ldb, Bp, ldb, n1, m, n1, 1.0, 0.0, GEMM_KHI_M);
        g.order = 1;
        GPX_TRY(gpx_gemm(s, 1, 0, g));
    }
    GPX_TRY(gpx_gemm(s, 1, 0,
                     mk(w.A + o11 + n1, ld, Bp, ldb, Bp + (size_t)n1 * ldb, ldb, n2, m, n1,
                        -1.0, 1.0, 0)));
    return trsm_rt_rec(s, w, off + n1, n2, Bp + (size_t)n1 * ldb, Tp + (size_t)n1 * ldb,
                       ldb, m);
}

int gpx_trsm_rt(hipStream_t s, const DenseWs &w, double *B, double *T, int ldb, int m)
{
    return trsm_rt_rec(s, w, 0, w.np, B, T, ldb, m);
}

int gpx_lauum(hipStream_t s, const DenseWs &w)
{
    // Kinv[i][j] = sum_{k >= max(i,j)} W[i][k] W[j][k], tiles with j >= i
    const int n = w.np, ld = w.ld;
    GemmArgs g = mk(w.W, ld, w.W, ld, w.Kinv, ld, n, n, n, 1.0, 0.0,
                    GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KLO_N |
                        (env_int("GPX_KREV", 0) ? GEMM_KREV : 0));
    g.order = env_int("GPX_ORD_LAUUM", 0);
    return gpx_gemm(s, 0, 1, g);
}
