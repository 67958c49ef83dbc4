// HBM-bound vector work of the ExactGP path: triangular solves with one right
// hand side, the log-marginal-likelihood scalars, the posterior reductions.
//
//   trsv_rt   a = R^-T (y - m)                       exact.py:55
//   trmv      alpha = R^-1 a as W a (W = R^-1)       exact.py:128
//   lz_terms  sum a^2, sum log R_ii, sum alpha       exact.py:119-121,141
//   posterior mu = m + V^T a, s2 = k** - colsum(V^2) exact.py:93-94
//
// (paths relative to /root/reference/pygp/inference/)

#include "gpx_internal.h"
#include <algorithm>

#define LB GPX_TILE

// ---- forward solve a = R^-T r --------------------------------------------------
// Block forward substitution over the diagonal blocks of the factorisation
// (chol.hip): a_k = W_kk^T r_k with the block's full inverse, r[k+1:] -= R[k, k+1:]^T
// a_k. Every step is a transposed mat-vec y[j] (op)= alpha * sum_i M[i][j] x[i], HBM-bound,
// done in two deterministic stages: 256x256 blocks write column partial sums
// (coalesced along j, x chunk in LDS), a second kernel adds the row chunks.
#define GV 256
__global__ __launch_bounds__(GV) void gemvt_partial_kernel(
    const double *__restrict__ M, int ld, int rows, int cols, int tri,
    const double *__restrict__ x, double *__restrict__ partial, long long sM, long long sx,
    long long sp)
{
    // member-batched launches: blockIdx.z = member, operands sM / sx / sp elements apart
    M += (long long)blockIdx.z * sM;
    x += (long long)blockIdx.z * sx;
    partial += (long long)blockIdx.z * sp;
    __shared__ double xs[GV];
    const int j = blockIdx.x * GV + threadIdx.x;
    const int i0 = blockIdx.y * GV;
    const int i1 = min(rows, i0 + GV);
    double acc = 0.0;
    // upper-triangular M (same origin for rows and columns): rows below the
    // last column of this block contribute nothing
    if (!tri || i0 <= blockIdx.x * GV + GV - 1) {
        if (i0 + threadIdx.x < i1) xs[threadIdx.x] = x[i0 + threadIdx.x];
        __syncthreads();
        if (j < cols) {
            const double *Mp = M + (size_t)i0 * ld + j;
            // below the 128-block of column j nothing was ever written
            int lim = i1 - i0;
            if (tri) lim = min(lim, (j / LB + 1) * LB - i0);
#pragma unroll 8
            for (int i = 0; i < lim; ++i) acc += Mp[(size_t)i * ld] * xs[i];
        }
    }
    if (j < cols) partial[(size_t)blockIdx.y * cols + j] = acc;
}

__global__ __launch_bounds__(GV) void gemvt_finish_kernel(const double *__restrict__ partial,
                                                          int nchunk, int cols, double alpha,
                                                          double beta, double *__restrict__ y,
                                                          long long sp, long long sy)
{
    partial += (long long)blockIdx.z * sp;
    y += (long long)blockIdx.z * sy;
    const int j = blockIdx.x * GV + threadIdx.x;
    if (j >= cols) return;
    double acc = 0.0;
    for (int c = 0; c < nchunk; ++c) acc += partial[(size_t)c * cols + j];
    y[j] = (beta != 0.0 ? beta * y[j] : 0.0) + alpha * acc;
}

// (batch members: matrices sM, vectors sv, partial sums sp elements apart)
static int gemvt(hipStream_t s, const double *M, int ld, int rows, int cols, bool tri,
                 const double *x, double alpha, double beta, double *y, double *partial,
                 int batch = 1, long long sM = 0, long long sv = 0, long long sp = 0)
{
    const int nchunk = (rows + GV - 1) / GV;
    dim3 grid((cols + GV - 1) / GV, nchunk, batch);
    hipLaunchKernelGGL(gemvt_partial_kernel, grid, dim3(GV), 0, s, M, ld, rows, cols,
                       tri ? 1 : 0, x, partial, sM, sv, sp);
    hipLaunchKernelGGL(gemvt_finish_kernel, dim3((cols + GV - 1) / GV, 1, batch), dim3(GV), 0, s,
                       partial, nchunk, cols, alpha, beta, y, sp, sv);
    GPX_HIP(hipGetLastError());
    return 0;
}

size_t gpx_trsv_scratch(int np) { return (size_t)((np + GV - 1) / GV) * np; }

int gpx_trsv_rt(hipStream_t s, const DenseWs &w, bool w_complete, double *r_scratch,
                double *a, double *partial, long long vstride)
{
    const int nb = w.batch;
    const long long sM = w.mstride, sv = vstride, sp = (long long)gpx_trsv_scratch(w.np);
    if (w_complete)        // a = W^T r in one sweep over the upper triangle
        return gemvt(s, w.W, w.ld, w.np, w.np, true, r_scratch, 1.0, 0.0, a, partial, nb, sM,
                     sv, sp);
    // block forward substitution with the inverses of the diagonal blocks (what a
    // value-only gpx_potrf leaves in W): a_k = W_kk^T r_k, r[k+1:] -= R[k, k+1:]^T a_k
    const GpxBlocks bl(w.np);
    const int ld = w.ld;
    for (int k = 0; k < bl.count; ++k) {
        const int ok = bl.off(k), nk = bl.len(k), o1 = ok + nk;
        const size_t okk = (size_t)ok * ld + ok;
        GPX_TRY(gemvt(s, w.W + okk, ld, nk, nk, true, r_scratch + ok, 1.0, 0.0, a + ok, partial,
                      nb, sM, sv, sp));
        if (o1 < w.np)
            GPX_TRY(gemvt(s, w.A + okk + nk, ld, nk, w.np - o1, false, a + ok, -1.0, 1.0,
                          r_scratch + o1, partial, nb, sM, sv, sp));
    }
    return 0;
}

// ---- out = W v, W upper triangular: one wave per row ------------------------
__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void trmv_upper_kernel(const double *__restrict__ W,
                                                         int ld, int np,
                                                         const double *__restrict__ v,
                                                         double *__restrict__ out,
                                                         long long sM, long long sv)
{
    W += (long long)blockIdx.z * sM;
    v += (long long)blockIdx.z * sv;
    out += (long long)blockIdx.z * sv;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= np) return;
    const double *Wr = W + (size_t)row * ld;
    // start at the 128-aligned column at or left of the diagonal: the leaf
    // blocks hold explicit zeros below the diagonal
    const int j0 = (row / LB) * LB;
    double acc = 0.0;
    for (int j = j0 + 2 * lane; j < np; j += 128) {
        const double2 wv = *reinterpret_cast<const double2 *>(Wr + j);
        const double2 vv = *reinterpret_cast<const double2 *>(v + j);
        acc += wv.x * vv.x + wv.y * vv.y;
    }
    acc = wave_sum64(acc);
    if (lane == 0) out[row] = acc;
}

int gpx_trmv_upper(hipStream_t s, const double *W, int ld, int np, const double *v,
                   double *out, int batch, long long mstride, long long vstride)
{
    hipLaunchKernelGGL(trmv_upper_kernel, dim3((np + 3) / 4, 1, batch), dim3(256), 0, s, W, ld,
                       np, v, out, mstride, vstride);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- scalars of the log marginal likelihood ----------------------------------
__global__ __launch_bounds__(1024) void lz_terms_kernel(const double *__restrict__ R,
                                                        int np, int n,
                                                        const double *__restrict__ a,
                                                        const double *__restrict__ alpha,
                                                        double *__restrict__ scalars,
                                                        long long sM, long long sv, int ss,
                                                        const int *__restrict__ info, int sa)
{
    // (sa: element stride of a -- 1, or the row stride of the matrix whose right-hand-side
    // column still holds it: value-only members of a group, no copy in between)
    R += (long long)blockIdx.x * sM;                     // blockIdx.x = member
    a += (long long)blockIdx.x * sv;
    if (alpha) alpha += (long long)blockIdx.x * sv;
    scalars += (long long)blockIdx.x * ss;
    __shared__ double red[3][16];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double ai = a[(size_t)i * sa];
        s0 += ai * ai;                                  // exact.py:119
        s1 += log(R[(size_t)i * np + i]);               // exact.py:121
        if (alpha) s2 += alpha[i];                      // exact.py:141
    }
    s0 = wave_sum64(s0);
    s1 = wave_sum64(s1);
    s2 = wave_sum64(s2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
        red[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int wv = 0; wv < 16; ++wv) t += red[threadIdx.x][wv];
        scalars[threadIdx.x] = t;
    }
    // the factorisation's status rides along in the fourth slot (one result copy, not two)
    if (threadIdx.x == 3 && info) scalars[3] = (double)info[blockIdx.x];
}

int gpx_lz_terms(hipStream_t s, const double *R, int np, int n, const double *a,
                 const double *alpha, double *scalars, int batch, long long mstride,
                 long long vstride, int sstride, const int *info, int astride)
{
    hipLaunchKernelGGL(lz_terms_kernel, dim3(batch), dim3(1024), 0, s, R, np, n, a, alpha,
                       scalars, mstride, vstride, sstride, info, astride);
    GPX_HIP(hipGetLastError());
    return 0;
}

__global__ void residual_kernel(const double *__restrict__ y, double mean, int n, int np,
                                double *__restrict__ r)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np) r[i] = i < n ? y[i] - mean : 0.0;       // exact.py:53
}

// Member-batched residuals (blockIdx.z = member, its mean from its parameter record).
// aug != null: the residual also goes into column `np` of the member's staging matrix, as
// the right-hand side of a whole-matrix panel launch (its 127 columns to the right zero).
__global__ void residual_members_kernel(const double *__restrict__ y,
                                        const MemberParams *__restrict__ mp, double mean1, int n,
                                        int np, double *__restrict__ r, long long vstride,
                                        double *__restrict__ aug, int ld, long long mstride,
                                        int *__restrict__ info0)
{
    // one thread per (row, pair of columns) of the 128-column strip: 64 threads a row
    // (mp == null: one model, its mean by value)
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int i = e >> 6, c = e & 63;
    // (info0: the members' status words, cleared here for the factorisation that follows --
    // no fill launch of its own)
    if (info0 && e == 0) info0[blockIdx.z] = 0;
    if (i >= np) return;
    const double v = i < n ? y[i] - (mp ? mp[blockIdx.z].mean : mean1) : 0.0;
    if (r && c == 0) r[(long long)blockIdx.z * vstride + i] = v;
    if (aug)
        reinterpret_cast<double2 *>(aug + (long long)blockIdx.z * mstride + (size_t)i * ld +
                                    np)[c] = make_double2(c == 0 ? v : 0.0, 0.0);
}

int gpx_residual_members(hipStream_t s, const double *y, const MemberBatch &mb, int n, int np,
                         double *r, double *aug, int ld, int *info_zero)
{
    hipLaunchKernelGGL(residual_members_kernel, dim3((np * 64 + 255) / 256, 1, mb.count),
                       dim3(256), 0, s, y, mb.params, 0.0, n, np, r, mb.vstride, aug, ld,
                       mb.mstride, info_zero);
    GPX_HIP(hipGetLastError());
    return 0;
}

// one model: r = y - mean (may be null) and the right-hand-side strip of its staging matrix
int gpx_residual_rhs(hipStream_t s, const double *y, double mean, int n, int np, double *r,
                     double *aug, int ld)
{
    hipLaunchKernelGGL(residual_members_kernel, dim3((np * 64 + 255) / 256, 1, 1), dim3(256), 0, s,
                       y, (const MemberParams *)nullptr, mean, n, np, r, 0LL, aug, ld, 0LL,
                       (int *)nullptr);
    GPX_HIP(hipGetLastError());
    return 0;
}

// column `col` of the members' matrices -> their vectors (a = R^-T r out of a whole-matrix
// launch with a right-hand side)
__global__ void column_out_kernel(const double *__restrict__ A, int ld, int col, int np,
                                  long long mstride, double *__restrict__ out, long long vstride)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np)
        out[(long long)blockIdx.z * vstride + i] =
            A[(long long)blockIdx.z * mstride + (size_t)i * ld + col];
}

int gpx_column_out(hipStream_t s, const double *A, int ld, int col, int np, double *out,
                   const MemberBatch &mb)
{
    hipLaunchKernelGGL(column_out_kernel, dim3((np + 255) / 256, 1, mb.count), dim3(256), 0, s, A,
                       ld, col, np, mb.mstride, out, mb.vstride);
    GPX_HIP(hipGetLastError());
    return 0;
}

int gpx_residual(hipStream_t s, const double *y, double mean, int n, int np, double *r)
{
    hipLaunchKernelGGL(residual_kernel, dim3((np + 255) / 256), dim3(256), 0, s, y, mean,
                       n, np, r);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- posterior reductions ----------------------------------------------------
#define PR_CHUNKS 256    // row chunks: enough workgroups also for a handful of test points
__global__ __launch_bounds__(256) void posterior_partial_kernel(
    const double *__restrict__ V, int ldv, int np, int m, const double *__restrict__ a,
    double *__restrict__ part, int nsplit, long long split_stride, long long sV, long long sa,
    long long spart)
{
    V += (long long)blockIdx.z * sV;                     // blockIdx.z = member
    a += (long long)blockIdx.z * sa;
    part += (long long)blockIdx.z * spart;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int rows = (np + (int)gridDim.y - 1) / (int)gridDim.y;
    const int i0 = chunk * rows, i1 = min(np, i0 + rows);
    if (j >= m) return;
    double smu = 0.0, ssq = 0.0;
    for (int i = i0; i < i1; ++i) {
        double v = V[(size_t)i * ldv + j];
        for (int sp = 1; sp < nsplit; ++sp) v += V[sp * split_stride + (size_t)i * ldv + j];
        smu += v * a[i];                                // exact.py:93
        ssq += v * v;                                   // exact.py:94
    }
    part[((size_t)chunk * 2 + 0) * m + j] = smu;
    part[((size_t)chunk * 2 + 1) * m + j] = ssq;
}

__global__ __launch_bounds__(256) void posterior_final_kernel(
    const double *__restrict__ part, int m, int chunks, double mean, double prior,
    double *__restrict__ mu, double *__restrict__ s2, const MemberParams *__restrict__ mp,
    long long spart, long long sout)
{
    if (mp) {                                            // member-batched: blockIdx.z
        mean = mp[blockIdx.z].mean;
        prior = mp[blockIdx.z].prior;
        part += (long long)blockIdx.z * spart;
        mu += (long long)blockIdx.z * sout;
        s2 += (long long)blockIdx.z * sout;
    }
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double smu = 0.0, ssq = 0.0;
    for (int c = 0; c < chunks; ++c) {
        smu += part[((size_t)c * 2 + 0) * m + j];
        ssq += part[((size_t)c * 2 + 1) * m + j];
    }
    mu[j] = mean + smu;
    s2[j] = prior - ssq;
}

// out[i][j] = sum_s P[s * stride + i * ld + j]: the partial products of a split-K
// launch, added in a fixed order
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ P,
                                                           int nsplit, long long stride,
                                                           long long count,
                                                           double *__restrict__ out,
                                                           long long sP, long long sout)
{
    P += (long long)blockIdx.z * sP;                     // blockIdx.z = member
    out += (long long)blockIdx.z * sout;
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (e >= count) return;
    double2 acc = *reinterpret_cast<const double2 *>(P + e);
    for (int sp = 1; sp < nsplit; ++sp) {
        const double2 v = *reinterpret_cast<const double2 *>(P + sp * stride + e);
        acc.x += v.x;
        acc.y += v.y;
    }
    *reinterpret_cast<double2 *>(out + e) = acc;
}

int gpx_sum_partials(hipStream_t s, const double *P, int nsplit, long long stride,
                     long long count, double *out, int batch, long long sP, long long sout)
{
    const long long pairs = (count + 1) / 2;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((pairs + 255) / 256), 1, batch),
                       dim3(256), 0, s, P, nsplit, stride, count, out, sP, sout);
    GPX_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void sub_partials_kernel(const double *__restrict__ P,
                                                           int nsplit, long long stride,
                                                           int rows, int cols,
                                                           double *__restrict__ C, int ldc)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * cols) return;
    double acc = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) acc += P[sp * stride + e];
    C[(size_t)(e / cols) * ldc + e % cols] -= acc;
}

int gpx_sub_partials(hipStream_t s, const double *P, int nsplit, long long stride, int rows,
                     int cols, double *C, int ldc)
{
    hipLaunchKernelGGL(sub_partials_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, s, P,
                       nsplit, stride, rows, cols, C, ldc);
    GPX_HIP(hipGetLastError());
    return 0;
}

// part: scratch of gpx_posterior_scratch(m) doubles
size_t gpx_posterior_scratch(int m) { return (size_t)2 * PR_CHUNKS * m; }

int gpx_posterior_reduce(hipStream_t s, const double *V, int ldv, int np, int m,
                          const double *a, double mean, double prior, double *part,
                          double *mu, double *s2, int nsplit, long long split_stride,
                          const MemberBatch *mb, long long vstride_panel)
{
    // (mb: the members' panels V lie vstride_panel, their vectors mb->vstride, their partial
    // sums gpx_posterior_scratch(m) and their outputs m doubles apart; mean and prior come
    // from their records)
    // row chunks of ~64 rows, at least 16: enough workgroups also for one test point
    const int chunks = std::min(PR_CHUNKS, std::max(16, np / 64));
    const int members = mb ? mb->count : 1;
    const long long spart = mb ? (long long)gpx_posterior_scratch(m) : 0;
    dim3 grid((m + 255) / 256, chunks, members);
    hipLaunchKernelGGL(posterior_partial_kernel, grid, dim3(256), 0, s, V, ldv, np, m, a,
                       part, nsplit, split_stride, mb ? vstride_panel : 0,
                       mb ? mb->vstride : 0, spart);
    hipLaunchKernelGGL(posterior_final_kernel, dim3((m + 255) / 256, 1, members), dim3(256), 0, s,
                       part, m, chunks, mean, prior, mu, s2, mb ? mb->params : nullptr, spart,
                       (long long)m);
    GPX_HIP(hipGetLastError());
    return 0;
}

// upper triangle copy with zeroed strict lower part (gp._R as the reference
// exposes it, exact.py:54) from the padded device factor
__global__ void copy_upper_kernel(const double *__restrict__ A, int np, int n,
                                  double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) out[(size_t)i * n + j] = j >= i ? A[(size_t)i * np + j] : 0.0;
}

int gpx_copy_upper(hipStream_t s, const double *A, int np, int n, double *out)
{
    hipLaunchKernelGGL(copy_upper_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, A, np,
                       n, out);
    GPX_HIP(hipGetLastError());
    return 0;
}

// full symmetric matrix from its upper triangle
__global__ void symmetrize_kernel(const double *__restrict__ A, int np, int n,
                                  double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n)
        out[(size_t)i * n + j] = j >= i ? A[(size_t)i * np + j] : A[(size_t)j * np + i];
}

int gpx_symmetrize(hipStream_t s, const double *A, int np, int n, double *out)
{
    hipLaunchKernelGGL(symmetrize_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, A, np,
                       n, out);
    GPX_HIP(hipGetLastError());
    return 0;
}
