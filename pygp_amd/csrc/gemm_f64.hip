// fp64 tile engine for gfx950: C = alpha * op(A) * op(B) + beta * C on
// v_mfma_f64_16x16x4_f64.
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 tile of C; each wave
// owns 64x64 = 4x4 MFMA tiles (16 accumulators of 4 f64 = 128 VGPRs). The K
// loop walks BK=16 slices: global -> registers (16-B loads, one slice ahead)
// -> LDS (double buffered, k-major image As[k][m], Bs[k][n], row stride 144
// doubles so that the two k-rows a 32-lane group touches sit on disjoint
// banks) -> one ds_read_b64 per fragment -> MFMA. A slice costs each wave
// 64 MFMAs, so LDS and global traffic are far off the critical path; what
// matters is that all four SIMDs always have an MFMA to issue.
//
// Triangular structure is exploited at tile granularity: per-tile k-ranges
// (GEMM_KLO_* / GEMM_KHI_*) skip slices that are structurally zero, and
// GEMM_UPPER_ONLY drops tiles below the diagonal. This is what lets the
// Cholesky / inverse drivers in chol.hip run SYRK, TRMM-like and LAUUM-like
// products through one kernel at ~N^3/3 flops each.
//
// Replaces (together with chol.hip): LAPACK dpotrf / dtrtrs / dpotrs reached
// from /root/reference/pygp/inference/exact.py:54-55,88,128-129.

#include "gpx_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define BM GPX_TILE
#define BN GPX_TILE
#define BK GPX_BK
#define LSTR 144                      // LDS row stride (doubles)
#define LDS_BYTES (2 * 2 * BK * LSTR * 8)

// ---- tile loaders -----------------------------------------------------------
// KMAJOR: the operand is stored with k as the slow index ([k][mn]); a slice is
// 16 rows of 128 contiguous doubles. Otherwise it is stored [mn][k]; a slice
// is 128 rows of 16 contiguous doubles (one 128-B line each).
template <bool KMAJOR>
__device__ __forceinline__ void load_slice(const double *__restrict__ P, int ld,
                                           int mn0, int k0, int tid,
                                           double2 (&r)[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int idx = tid + 256 * c;
        if (KMAJOR) {
            const int row = idx >> 6, c2 = idx & 63;
            r[c] = *reinterpret_cast<const double2 *>(
                P + (size_t)(k0 + row) * ld + mn0 + 2 * c2);
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            r[c] = *reinterpret_cast<const double2 *>(
                P + (size_t)(mn0 + row) * ld + k0 + 2 * k2);
        }
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_slice(double *__restrict__ S, int tid,
                                            const double2 (&r)[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int idx = tid + 256 * c;
        if (KMAJOR) {
            const int row = idx >> 6, c2 = idx & 63;
            *reinterpret_cast<double2 *>(S + row * LSTR + 2 * c2) = r[c];
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            S[(2 * k2) * LSTR + row] = r[c].x;
            S[(2 * k2 + 1) * LSTR + row] = r[c].y;
        }
    }
}

// the 64 MFMAs one wave issues for one BK=16 slice: 4 k-steps x (4x4 tiles)
__device__ __forceinline__ void mfma_slice(const double *__restrict__ ap,
                                           const double *__restrict__ bp,
                                           v4d (&acc)[4][4])
{
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = ap[ks * 4 * LSTR + t * 16];
            b[t] = bp[ks * 4 * LSTR + t * 16];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j],
                                                                 0, 0, 0);
    }
}

template <int TA, int TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if ((g.flags & GEMM_UPPER_ONLY) && n0 + BN <= m0) return;
    int klo = 0, khi = g.K;
    if (g.flags & GEMM_KLO_M) klo = max(klo, m0);
    if (g.flags & GEMM_KHI_M) khi = min(khi, m0 + BM);
    if (g.flags & GEMM_KLO_N) klo = max(klo, n0);
    if (g.flags & GEMM_KHI_N) khi = min(khi, n0 + BN);

    const double *__restrict__ A = g.A + (long long)blockIdx.z * g.strideA;
    const double *__restrict__ B = g.B + (long long)blockIdx.z * g.strideB;
    double *__restrict__ C = g.C + (long long)blockIdx.z * g.strideC;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;

    // op(A)[m][k]: TA == 0 -> A[m*lda + k] (mn-major), TA == 1 -> A[k*lda + m]
    // op(B)[k][n]: TB == 0 -> B[k*ldb + n] (k-major),  TB == 1 -> B[n*ldb + k]
    constexpr bool AKM = (TA == 1);
    constexpr bool BKM = (TB == 0);

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int nslice = (khi - klo) / BK;
    if (nslice > 0) {
        double2 ra[4], rb[4];
        load_slice<AKM>(A, g.lda, m0, klo, tid, ra);
        load_slice<BKM>(B, g.ldb, n0, klo, tid, rb);
        store_slice<AKM>(smem, tid, ra);
        store_slice<BKM>(smem + 2 * BK * LSTR, tid, rb);
        __syncthreads();

        // steady state: prefetch slice s+1 into registers (unconditionally, so
        // that ra/rb stay in VGPRs -- a conditional prefetch sends them to
        // scratch), run the 64 MFMAs of slice s, then publish s+1 to LDS
        const double *ap0 = smem + lk * LSTR + wm * 64 + lr;
        const double *bp0 = smem + 2 * BK * LSTR + lk * LSTR + wn * 64 + lr;
        for (int s = 0; s + 1 < nslice; ++s) {
            const int cur = s & 1;
            const int k0 = klo + (s + 1) * BK;
            load_slice<AKM>(A, g.lda, m0, k0, tid, ra);
            load_slice<BKM>(B, g.ldb, n0, k0, tid, rb);
            mfma_slice(ap0 + cur * BK * LSTR, bp0 + cur * BK * LSTR, acc);
            const int nxt = cur ^ 1;
            store_slice<AKM>(smem + nxt * BK * LSTR, tid, ra);
            store_slice<BKM>(smem + 2 * BK * LSTR + nxt * BK * LSTR, tid, rb);
            __syncthreads();
        }
        {
            const int cur = (nslice - 1) & 1;
            mfma_slice(ap0 + cur * BK * LSTR, bp0 + cur * BK * LSTR, acc);
        }
    }

    // epilogue. f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + lk + 4 * r;
                const int col = n0 + wn * 64 + j * 16 + lr;
                double *p = C + (size_t)row * g.ldc + col;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * (*p);
                *p = v;
            }
}

template <int TA, int TB>
static int launch(hipStream_t s, const GemmArgs &g)
{
    dim3 grid(g.N / BN, g.M / BM, g.batch > 0 ? g.batch : 1);
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB>), grid, dim3(256), LDS_BYTES, s, g);
    GPX_HIP(hipGetLastError());
    return 0;
}

template <int TA, int TB>
static int set_attr()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f64_kernel<TA, TB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    return 0;
}

int gpx_gemm_init()
{
    GPX_TRY((set_attr<0, 0>()));
    GPX_TRY((set_attr<0, 1>()));
    GPX_TRY((set_attr<1, 0>()));
    GPX_TRY((set_attr<1, 1>()));
    return 0;
}

int gpx_gemm(hipStream_t s, int ta, int tb, const GemmArgs &g)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.M % BM || g.N % BN || g.K % BK || g.lda % 2 || g.ldb % 2) {
        gpx_set_error("gpx_gemm: unpadded operands M=%d N=%d K=%d lda=%d ldb=%d",
                      g.M, g.N, g.K, g.lda, g.ldb);
        return -1;
    }
    if (ta == 0 && tb == 0) return launch<0, 0>(s, g);
    if (ta == 0 && tb == 1) return launch<0, 1>(s, g);
    if (ta == 1 && tb == 0) return launch<1, 0>(s, g);
    return launch<1, 1>(s, g);
}
