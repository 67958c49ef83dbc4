// Blocked Cholesky, triangular inverse and symmetric inverse on the fp64 tile
// engine (gemm_f64.hip). Everything works on the UPPER factor R of the
// reference (R^T R = K + sn2 I, scipy.linalg.cholesky at
// /root/reference/pygp/inference/exact.py:54) in row-major storage.
//
//   potrf  recursive right-looking: factor the leading half, solve the row panel
//          R12 = R11^-T A12, SYRK the trailing half on MFMA (upper tiles only),
//          recurse. The 128x128 diagonal leaves are factored by one workgroup
//          with the block held in registers (8x8 cyclic tile per thread) and
//          one LDS row broadcast per pivot; the same kernel inverts the leaf.
//          The row-panel solve never substitutes: at the leaves it multiplies
//          by the explicit leaf inverse (a K=128 MFMA GEMM), above them it is
//          GEMM updates only.
//   trtri  W = R^-1 from the leaf inverses: W12 = -W11 (R12 W22), two
//          triangle-aware GEMMs per node (exact.py:129 needs K^-1).
//   lauum  Kinv = W W^T, upper tiles only, one launch with per-tile k ranges.
//
// potrf + trtri + lauum = N^3/3 + N^3/3 + N^3/3 flops, against the 7N^3/3 of
// the reference's cho_solve(R, eye(N)) route (exact.py:129).

#include "gpx_internal.h"

#define LB GPX_TILE                    // leaf order
#define LSTRIDE (LB + 1)
#define LEAF_LDS ((LB * LSTRIDE + 2 * LB + LB) * 8)

// ---- leaf: R = chol(A11) and W = R^-1 in one workgroup ----------------------
__global__ __launch_bounds__(256) void potrf_leaf_kernel(double *__restrict__ A, int lda,
                                                         double *__restrict__ W, int ldw,
                                                         int *__restrict__ info, int goff)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *Rs = reinterpret_cast<double *>(smem_raw);      // [LB][LSTRIDE]
    double *rowbuf = Rs + LB * LSTRIDE;                      // [2][LB]
    double *dinv = rowbuf + 2 * LB;                          // [LB]

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int wave = tid >> 6, lane = tid & 63;

    // cyclic 8x8 register tile: rows ty + 16a, cols tx + 16b
    double s[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            s[a][b] = A[(size_t)(ty + 16 * a) * lda + tx + 16 * b];

    // ---- factorisation: 128 pivots, one barrier each ----
#pragma unroll
    for (int ak = 0; ak < 8; ++ak) {
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 16 * ak + kk;
            double *rb = rowbuf + (k & 1) * LB;
            if (wave == (kk >> 2)) {
                // pivot lives in lane (ty = kk, tx = kk) of this wave
                const double d = __shfl(s[ak][ak], ((kk & 3) << 4) | kk, 64);
                const double sq = sqrt(d);
                const double rinv = 1.0 / sq;
                if (!(d > 0.0) && lane == 0 && *info == 0) *info = goff + k + 1;
                if (ty == kk) {
                    if (tx == 0) dinv[k] = rinv;
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const int j = tx + 16 * b;
                        double v = 0.0;
                        if (b >= ak) v = (j > k) ? s[ak][b] * rinv : (j == k ? sq : 0.0);
                        A[(size_t)k * lda + j] = v;
                        Rs[k * LSTRIDE + j] = v;
                        rb[j] = (j > k) ? v : 0.0;
                    }
                }
            }
            __syncthreads();
            double ri[8], cj[8];
#pragma unroll
            for (int a = 0; a < 8; ++a)
                if (a >= ak) {
                    ri[a] = rb[ty + 16 * a];
                    cj[a] = rb[tx + 16 * a];
                }
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (a >= ak && b >= ak) s[a][b] -= ri[a] * cj[b];
        }
    }

    // ---- inverse: solve R W = I bottom-up with rank-1 updates ----
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            s[a][b] = (ty + 16 * a == tx + 16 * b) ? 1.0 : 0.0;

#pragma unroll
    for (int ai = 7; ai >= 0; --ai) {
        for (int ii = 15; ii >= 0; --ii) {
            const int i = 16 * ai + ii;
            double *rb = rowbuf + (i & 1) * LB;
            if (ty == ii) {
                const double di = dinv[i];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int j = tx + 16 * b;
                    double v = 0.0;
                    if (b >= ai && j >= i) v = s[ai][b] * di;
                    W[(size_t)i * ldw + j] = v;
                    rb[j] = v;
                }
            }
            __syncthreads();
            double c[8], wj[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                if (a <= ai) {
                    const int r = ty + 16 * a;
                    c[a] = (r < i) ? Rs[r * LSTRIDE + i] : 0.0;
                }
                if (a >= ai) wj[a] = rb[tx + 16 * a];
            }
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (a <= ai && b >= ai) s[a][b] -= c[a] * wj[b];
        }
    }
}

int gpx_leaf_init()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_leaf_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LEAF_LDS));
    return 0;
}

int gpx_potrf_leaf(hipStream_t s, double *Ablk, int lda, double *Wblk, int ldw,
                   int *info, int goff)
{
    hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(256), LEAF_LDS, s, Ablk, lda,
                       Wblk, ldw, info, goff);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- recursion helpers ------------------------------------------------------
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
    return g;
}

// X = R[r0:r0+n, r0:r0+n]^-T B, B = n x m at Bp (ld ldb), in place.
static int trsm_rt_rec(hipStream_t s, const DenseWs &w, int r0, int n, double *Bp,
                       int ldb, int m)
{
    const int ld = w.np;
    if (n == LB) {
        // X = W_leaf^T B : op(A)[i][k] = W[k][i], lower triangular
        return gpx_gemm(s, 1, 0,
                        mk(w.W + (size_t)r0 * ld + r0, ld, Bp, ldb, Bp, ldb, LB, m, LB,
                           1.0, 0.0, 0));
    }
    const int n1 = split(n), n2 = n - n1;
    GPX_TRY(trsm_rt_rec(s, w, r0, n1, Bp, ldb, m));
    // B2 -= R12^T X1
    GPX_TRY(gpx_gemm(s, 1, 0,
                     mk(w.A + (size_t)r0 * ld + r0 + n1, ld, Bp, ldb,
                        Bp + (size_t)n1 * ldb, ldb, n2, m, n1, -1.0, 1.0, 0)));
    return trsm_rt_rec(s, w, r0 + n1, n2, Bp + (size_t)n1 * ldb, ldb, m);
}

static int potrf_rec(hipStream_t s, const DenseWs &w, int off, int n)
{
    const int ld = w.np;
    double *Aoo = w.A + (size_t)off * ld + off;
    if (n == LB)
        return gpx_potrf_leaf(s, Aoo, ld, w.W + (size_t)off * ld + off, ld, w.info, off);
    const int n1 = split(n), n2 = n - n1;
    GPX_TRY(potrf_rec(s, w, off, n1));
    double *A12 = Aoo + n1;
    GPX_TRY(trsm_rt_rec(s, w, off, n1, A12, ld, n2));
    // A22 -= R12^T R12, upper tiles only
    GPX_TRY(gpx_gemm(s, 1, 0,
                     mk(A12, ld, A12, ld, Aoo + (size_t)n1 * ld + n1, ld, n2, n2, n1,
                        -1.0, 1.0, GEMM_UPPER_ONLY)));
    return potrf_rec(s, w, off + n1, n2);
}

int gpx_potrf(hipStream_t s, const DenseWs &w)
{
    if (w.np % LB) {
        gpx_set_error("potrf: order %d not padded to %d", w.np, LB);
        return -1;
    }
    return potrf_rec(s, w, 0, w.np);
}

int gpx_trsm_rt(hipStream_t s, const DenseWs &w, double *B, int ldb, int m)
{
    return trsm_rt_rec(s, w, 0, w.np, B, ldb, m);
}

// W[off:off+n] = R[off:off+n]^-1 given the leaf inverses; Kinv's (1,2) block
// of each node is the temporary T = R12 W22.
static int trtri_rec(hipStream_t s, const DenseWs &w, int off, int n)
{
    if (n == LB) return 0;
    const int ld = w.np;
    const int n1 = split(n), n2 = n - n1;
    GPX_TRY(trtri_rec(s, w, off, n1));
    GPX_TRY(trtri_rec(s, w, off + n1, n2));
    const size_t o11 = (size_t)off * ld + off, o12 = o11 + n1,
                 o22 = (size_t)(off + n1) * ld + off + n1;
    // T = R12 W22 : op(B) = W22 upper -> k <= column tile
    GPX_TRY(gpx_gemm(s, 0, 0,
                     mk(w.A + o12, ld, w.W + o22, ld, w.Kinv + o12, ld, n1, n2, n2, 1.0,
                        0.0, GEMM_KHI_N)));
    // W12 = -W11 T : op(A) = W11 upper -> k >= row tile
    return gpx_gemm(s, 0, 0,
                    mk(w.W + o11, ld, w.Kinv + o12, ld, w.W + o12, ld, n1, n2, n1, -1.0,
                       0.0, GEMM_KLO_M));
}

int gpx_trtri(hipStream_t s, const DenseWs &w) { return trtri_rec(s, w, 0, w.np); }

int gpx_lauum(hipStream_t s, const DenseWs &w)
{
    // Kinv[i][j] = sum_{k >= max(i,j)} W[i][k] W[j][k], tiles with j >= i
    const int n = w.np;
    return gpx_gemm(s, 0, 1,
                    mk(w.W, n, w.W, n, w.Kinv, n, n, n, n, 1.0, 0.0,
                       GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KLO_N));
}
