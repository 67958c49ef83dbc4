// Building blocks of the fp64 tile engine shared by gemm_f64.hip (whole-launch
// GEMMs) and panel.hip (tile tasks of the diagonal-panel kernel): workgroup
// geometry, the global -> register -> LDS slice loaders and the per-slice MFMA
// sequence. See gemm_f64.hip for the layout description.
#pragma once
#include "gpx_internal.h"

#ifndef GPX_V4D
#define GPX_V4D
typedef double v4d __attribute__((ext_vector_type(4)));
#endif

#define BK GPX_BK
#define MNSTR 18                      // row stride of an mn-major LDS image

// Workgroup geometry: TILE x TILE of C, WM x WN waves, each wave
// (TILE/WM) x (TILE/WN) = WTM x WTN MFMA tiles of 16x16.
template <int TILE_, int WM_, int WN_, bool DEEP_ = false> struct Geo {
    static constexpr int TILE = TILE_, WM = WM_, WN = WN_;
    static constexpr bool DEEP = DEEP_;                    // prefetch two slices ahead
    static constexpr int NTH = 64 * WM * WN;
    // HIP's second __launch_bounds__ argument is waves per SIMD (not blocks per
    // CU). The 8-wave shapes ask for 4 = two workgroups per CU, which holds the
    // kernel to 128 VGPRs: at 134 (NN, NT) only one workgroup fits a CU and its
    // two waves per SIMD stall together at every barrier -- NN 68.0 -> 72.3,
    // NT 67.0 -> 72.8, LAUUM 66.4 -> 70.6 TFLOP/s (tools/gemm_cfg6.txt)
    static constexpr int MINW = (WM_ * WN_ >= 8) ? 4 : 2;
    static constexpr int WTM = TILE / WM / 16, WTN = TILE / WN / 16;
    static constexpr int KSTR = TILE + 16;                 // k-major row stride
    static constexpr int OPER = (BK * KSTR > TILE * MNSTR) ? BK * KSTR : TILE * MNSTR;
    static constexpr int LDS_BYTES = 2 * 2 * OPER * 8;     // 2 operands x 2 buffers
    static constexpr int NLOAD = TILE * BK / 2 / NTH;      // double2 per thread
    static_assert(NLOAD >= 1 && NLOAD * NTH * 2 == TILE * BK, "loader shape");
};

// ---- slice loaders ----------------------------------------------------------
template <int N> struct Regs { double2 v[N]; };

template <typename G, bool KMAJOR>
__device__ __forceinline__ Regs<G::NLOAD> load_slice(const double *__restrict__ P, int ld,
                                                     int mn0, int k0, int tid)
{
    Regs<G::NLOAD> out;
    double2 (&r)[G::NLOAD] = out.v;
#pragma unroll
    for (int c = 0; c < G::NLOAD; ++c) {
        const int idx = tid + G::NTH * c;
        if (KMAJOR) {
            const int row = idx / (G::TILE / 2), c2 = idx % (G::TILE / 2);
            r[c] = *reinterpret_cast<const double2 *>(P + (size_t)(k0 + row) * ld + mn0 +
                                                      2 * c2);
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            r[c] = *reinterpret_cast<const double2 *>(P + (size_t)(mn0 + row) * ld + k0 +
                                                      2 * k2);
        }
    }
    return out;
}

template <typename G, bool KMAJOR>
__device__ __forceinline__ void store_slice(double *__restrict__ S, int tid,
                                            const Regs<G::NLOAD> &in)
{
    const double2 (&r)[G::NLOAD] = in.v;
#pragma unroll
    for (int c = 0; c < G::NLOAD; ++c) {
        const int idx = tid + G::NTH * c;
        if (KMAJOR) {
            const int row = idx / (G::TILE / 2), c2 = idx % (G::TILE / 2);
            *reinterpret_cast<double2 *>(S + row * G::KSTR + 2 * c2) = r[c];
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            *reinterpret_cast<double2 *>(S + row * MNSTR + 2 * k2) = r[c];
        }
    }
}

// the MFMAs one wave issues for one BK=16 slice: 4 k-steps x (WTM x WTN tiles).
// ap/bp point at this lane's first fragment element; AK/BKS and AT/BT are the
// LDS distances of one k-step and of one 16-wide MFMA tile per operand.
template <int WTM, int WTN, int AK, int AT, int BKS, int BT>
__device__ __forceinline__ void mfma_slice(const double *__restrict__ ap,
                                           const double *__restrict__ bp,
                                           v4d (&acc)[WTM][WTN])
{
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
        double a[WTM], b[WTN];
#pragma unroll
        for (int t = 0; t < WTM; ++t) a[t] = ap[ks * AK + t * AT];
#pragma unroll
        for (int t = 0; t < WTN; ++t) b[t] = bp[ks * BKS + t * BT];
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j],
                                                                 0, 0, 0);
    }
}

