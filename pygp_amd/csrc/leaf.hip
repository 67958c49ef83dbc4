// 128x128 diagonal leaf of the blocked Cholesky: R = chol(A) (upper) and
// W = R^-1, one workgroup of 4 waves, the block resident in LDS.
//
// The first version of this kernel walked 256 pivots with one workgroup barrier
// each and was bound by that latency chain (93 us). Here the work is blocked by 16:
//
//   per 16-row panel p
//     A(p)  wave 0 factors the 16x16 diagonal block entirely in registers: one
//           column per lane, pivots and multipliers move between lanes with
//           v_readlane (no LDS, no barrier). The same row operations applied to
//           the identity give Y = L^-1 = U^-T for free.
//     B(p)  row panel  R[p][q] = Y * A[p][q]            (MFMA 16x16x4, 4 per block)
//     C(p)  trailing   A[q][r] -= R[p][q]^T R[p][r]     (MFMA, 4 per block)
//           wave 0 only updates the next diagonal block and goes straight to
//           A(p+1) while waves 1-3 finish C(p): the serial chain of the leaf is
//           8 x (A + B), everything else hides behind it.
//   inverse: W = R^-1 from the 16x16 diagonal inverses by doubling (16 -> 32 ->
//           64): T = R12 W22 stays in MFMA accumulators, which are already in
//           the B-operand layout of the next product W12 = -W11 T (f64 MFMA:
//           D row = (lane>>4) + 4*reg = B's k index for k-step reg).
//
// Replaces the diagonal-block part of LAPACK dpotrf/dtrtri reached from
// /root/reference/pygp/inference/exact.py:54,129.

#include "gpx_internal.h"
#include <cstdlib>

#include "leaf_dev.h"

__global__ __launch_bounds__(256) void potrf_leaf2_kernel(double *__restrict__ A, int lda,
                                                          double *__restrict__ W, int ldw,
                                                          int *__restrict__ info, int goff,
                                                          int skip, long long mstride)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // blockIdx.x = member of a member-batched factorisation (its own info word)
    const long long mo = (long long)blockIdx.x * mstride;
    leaf2_run<false>(A + mo, lda, W + mo, ldw, info + blockIdx.x, goff, skip, smem_raw);
}

int gpx_leaf2_init()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&potrf_leaf2_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LEAF2_LDS));
    return 0;
}

int gpx_potrf_leaf2(hipStream_t s, double *Ablk, int lda, double *Wblk, int ldw, int *info,
                    int goff, int batch, long long mstride)
{
    static const int skip = [] {
        const char *e = getenv("GPX_LEAF_SKIP");
        const char *m = getenv("GPX_LEAF_MFMA");           // 0: register factorisation of the
        return (e ? atoi(e) : 0) | (m && !atoi(m) ? 32 : 0);   // 16 x 16 blocks (rounds 1-2)
    }();
    hipLaunchKernelGGL(potrf_leaf2_kernel, dim3(batch > 1 ? batch : 1), dim3(256), LEAF2_LDS, s,
                       Ablk, lda, Wblk, ldw, info, goff, skip, mstride);
    GPX_HIP(hipGetLastError());
    return 0;
}
