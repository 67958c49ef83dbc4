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

#define LB GPX_TILE

// ---- forward solve, one launch per 128-block --------------------------------
// Every workgroup recomputes x_J = W_JJ^T r_J (128x128 mat-vec out of L2) into
// LDS; block 0 stores it to `a`; then each workgroup applies the rank-128
// update r[j] -= sum_k R[J*128 + k][j] x_J[k] to its 256 columns right of the
// block (coalesced along j).
__global__ __launch_bounds__(256) void trsv_rt_step_kernel(
    const double *__restrict__ R, const double *__restrict__ W, int ld, int np, int J,
    double *__restrict__ r, double *__restrict__ a)
{
    __shared__ double x[LB];
    __shared__ double rj[LB];
    const int tid = threadIdx.x;
    const int c0 = J * LB;
    if (tid < LB) rj[tid] = r[c0 + tid];
    __syncthreads();
    {
        // two threads per output: halves of the k range
        const int t = tid & (LB - 1), half = tid >> 7;
        const double *Wp = W + (size_t)(c0 + half * 64) * ld + c0 + t;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) acc += Wp[(size_t)k * ld] * rj[half * 64 + k];
        if (half) x[t] = acc;
        __syncthreads();
        if (!half) x[t] += acc;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid < LB) a[c0 + tid] = x[tid];
    const int j = c0 + LB + blockIdx.x * 256 + tid;
    if (j < np) {
        const double *Rp = R + (size_t)c0 * ld + j;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < LB; ++k) acc += Rp[(size_t)k * ld] * x[k];
        r[j] -= acc;
    }
}

int gpx_trsv_rt(hipStream_t s, const DenseWs &w, double *r_scratch, double *a)
{
    const int nb = w.np / LB;
    for (int J = 0; J < nb; ++J) {
        const int rest = w.np - (J + 1) * LB;
        const int blocks = rest > 0 ? (rest + 255) / 256 : 1;
        hipLaunchKernelGGL(trsv_rt_step_kernel, dim3(blocks), dim3(256), 0, s, w.A, w.W,
                           w.ld, w.np, J, r_scratch, a);
    }
    GPX_HIP(hipGetLastError());
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
                                                         double *__restrict__ out)
{
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
                   double *out)
{
    hipLaunchKernelGGL(trmv_upper_kernel, dim3((np + 3) / 4), dim3(256), 0, s, W, ld, np, v,
                       out);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- scalars of the log marginal likelihood ----------------------------------
__global__ __launch_bounds__(1024) void lz_terms_kernel(const double *__restrict__ R,
                                                        int np, int n,
                                                        const double *__restrict__ a,
                                                        const double *__restrict__ alpha,
                                                        double *__restrict__ scalars)
{
    __shared__ double red[3][16];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double ai = a[i];
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
}

int gpx_lz_terms(hipStream_t s, const double *R, int np, int n, const double *a,
                 const double *alpha, double *scalars)
{
    hipLaunchKernelGGL(lz_terms_kernel, dim3(1), dim3(1024), 0, s, R, np, n, a, alpha,
                       scalars);
    GPX_HIP(hipGetLastError());
    return 0;
}

__global__ void residual_kernel(const double *__restrict__ y, double mean, int n, int np,
                                double *__restrict__ r)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np) r[i] = i < n ? y[i] - mean : 0.0;       // exact.py:53
}

int gpx_residual(hipStream_t s, const double *y, double mean, int n, int np, double *r)
{
    hipLaunchKernelGGL(residual_kernel, dim3((np + 255) / 256), dim3(256), 0, s, y, mean,
                       n, np, r);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- posterior reductions ----------------------------------------------------
#define PR_CHUNKS 32
__global__ __launch_bounds__(256) void posterior_partial_kernel(
    const double *__restrict__ V, int ldv, int np, int m, const double *__restrict__ a,
    double *__restrict__ part)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int rows = (np + PR_CHUNKS - 1) / PR_CHUNKS;
    const int i0 = chunk * rows, i1 = min(np, i0 + rows);
    if (j >= m) return;
    double smu = 0.0, ssq = 0.0;
    for (int i = i0; i < i1; ++i) {
        const double v = V[(size_t)i * ldv + j];
        smu += v * a[i];                                // exact.py:93
        ssq += v * v;                                   // exact.py:94
    }
    part[((size_t)chunk * 2 + 0) * m + j] = smu;
    part[((size_t)chunk * 2 + 1) * m + j] = ssq;
}

__global__ __launch_bounds__(256) void posterior_final_kernel(
    const double *__restrict__ part, int m, double mean, double prior,
    double *__restrict__ mu, double *__restrict__ s2)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double smu = 0.0, ssq = 0.0;
    for (int c = 0; c < PR_CHUNKS; ++c) {
        smu += part[((size_t)c * 2 + 0) * m + j];
        ssq += part[((size_t)c * 2 + 1) * m + j];
    }
    mu[j] = mean + smu;
    s2[j] = prior - ssq;
}

// part: scratch of gpx_posterior_scratch(m) doubles
size_t gpx_posterior_scratch(int m) { return (size_t)2 * PR_CHUNKS * m; }

int gpx_posterior_reduce(hipStream_t s, const double *V, int ldv, int np, int m,
                          const double *a, double mean, double prior, double *part,
                          double *mu, double *s2)
{
    dim3 grid((m + 255) / 256, PR_CHUNKS);
    hipLaunchKernelGGL(posterior_partial_kernel, grid, dim3(256), 0, s, V, ldv, np, m, a,
                       part);
    hipLaunchKernelGGL(posterior_final_kernel, dim3((m + 255) / 256), dim3(256), 0, s,
                       part, m, mean, prior, mu, s2);
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
