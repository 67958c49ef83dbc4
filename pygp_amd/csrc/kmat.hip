// Pairwise kernel evaluation on gfx950 (SE / Matern-1,3,5 / Periodic, ARD or
// iso lengthscales, sums of those), fp64 and fp32.
//
//   kbuild      K(X1, X2) tiles: scaled input blocks staged in LDS (k-major,
//               conflict-free 32-B reads), direct-difference squared distance
//               (exact zeros on the diagonal like scipy's cdist), transcendental
//               epilogue, 32-B coalesced row stores. Never materialises D^2.
//               Replaces _distances.py:17-41 + se.py:53-55 / matern.py:69-74 /
//               periodic.py:53-59 / _combo.py:106-108 and the "+ sn2*eye" of
//               inference/exact.py:52.
//   kgrad       all hyper-gradient slices for the Kernel.grad API
//               (se.py:57-66, matern.py:76-90, periodic.py:61-74).
//   trace_grad  the fused replacement of exact.py:130-138: streams the upper
//               triangle of K^-1 once, recomputes K and the per-dimension
//               squared differences from LDS-staged inputs, and accumulates
//               tr(Q) and sum(Q o dK_h) for every hyperparameter with wavefront
//               reductions. No N x N temporaries (the reference makes ~4 per
//               hyperparameter).
//
// All paths are relative to /root/reference/pygp/.

#include "gpx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

#define KT 64                  // output tile edge of kbuild / trace_grad
#ifndef GPX_TRACE_AHEAD
#define GPX_TRACE_AHEAD 0
#endif
#ifndef GPX_TRACE_UNROLL
#define GPX_TRACE_UNROLL 4
#endif

// ---- host: flatten a gpx_kspec tree -----------------------------------------
// The tree of sums and products (_combo.py:103-146 accepts any nesting) is expanded
// into a sum of products of primitive kernels: a product distributes over the sums
// among its factors. A primitive that the expansion repeats -- C in (A + B) C = A C +
// B C -- keeps ONE set of hyperparameters: every copy points at the same slots of the
// kernel's hyper vector (hoff) and the later copies are marked dup, so that their
// gradient contributions add up.
static int fill_leaf(const gpx_kspec *k, int64_t d, int hoff, KPart *out)
{
    if (!k->hyper) {
        gpx_set_error("kspec: null hyper");
        return -1;
    }
    KPart &p = *out;
    p.kind = k->kind;
    p.iso = k->iso;
    p.hoff = hoff;
    p.group = 0;
    p.dup = 0;
    p.two_logsf = k->hyper[0] * 2;
    p.sf2 = exp(k->hyper[0] * 2);
    p.ell = 1.0;
    p.pi_over_p = 0.0;
    p.period = 1.0;
    p.alpha = 1.0;
    for (int i = 0; i < GPX_MAX_DIM; ++i) p.scale[i] = 1.0;
    switch (k->kind) {
    case GPX_SE:
    case GPX_MATERN1:
    case GPX_MATERN3:
    case GPX_MATERN5: {
        const int nu = k->kind == GPX_SE ? 0
                       : (k->kind == GPX_MATERN1 ? 1 : (k->kind == GPX_MATERN3 ? 3 : 5));
        const int nell = k->iso ? 1 : (int)d;
        if (k->nhyper != 1 + nell || k->ndim != d) {
            gpx_set_error("kspec: kernel has ndim=%d nhyper=%d but data has d=%lld",
                          k->ndim, k->nhyper, (long long)d);
            return -1;
        }
        p.nhyper = 1 + nell;
        for (int i = 0; i < d; ++i) {
            double ell = exp(k->hyper[1 + (k->iso ? 0 : i)]);
            p.scale[i] = nu ? ell / sqrt((double)nu) : ell;   // matern.py:70
        }
        break;
    }
    case GPX_RQ: {                                     // rq.py:22-52
        const int nell = k->iso ? 1 : (int)d;
        if (k->nhyper != 2 + nell || k->ndim != d) {
            gpx_set_error("kspec: RQ kernel has ndim=%d nhyper=%d but data has d=%lld",
                          k->ndim, k->nhyper, (long long)d);
            return -1;
        }
        p.nhyper = 2 + nell;
        for (int i = 0; i < d; ++i) p.scale[i] = exp(k->hyper[1 + (k->iso ? 0 : i)]);
        p.alpha = exp(k->hyper[1 + nell]);
        break;
    }
    case GPX_PERIODIC:
        if (k->nhyper != 3) {
            gpx_set_error("kspec: periodic kernel needs 3 hypers");
            return -1;
        }
        p.nhyper = 3;
        p.ell = exp(k->hyper[1]);
        p.pi_over_p = M_PI / exp(k->hyper[2]);
        p.period = exp(k->hyper[2]);
        break;
    default:
        gpx_set_error("kspec: unknown kind %d", k->kind);
        return -1;
    }
    return 0;
}

// terms of the expansion of node k: each term lists leaf indices (into `leaves`)
typedef std::vector<std::vector<int>> Terms;
static int expand(const gpx_kspec *k, int64_t d, std::vector<KPart> &leaves, int *nhyper,
                  Terms *out)
{
    out->clear();
    if (k->kind == GPX_SUM || k->kind == GPX_PRODUCT) {
        if (k->nparts <= 0 || !k->parts) {
            gpx_set_error("kspec: empty sum / product");
            return -1;
        }
        if (k->kind == GPX_PRODUCT) out->push_back({});
        for (int i = 0; i < k->nparts; ++i) {
            Terms sub;
            GPX_TRY(expand(&k->parts[i], d, leaves, nhyper, &sub));
            if (k->kind == GPX_SUM) {
                out->insert(out->end(), sub.begin(), sub.end());
            } else {
                Terms next;
                for (const auto &t : *out)
                    for (const auto &u : sub) {
                        next.push_back(t);
                        next.back().insert(next.back().end(), u.begin(), u.end());
                    }
                out->swap(next);
            }
            size_t factors = 0;
            for (const auto &t : *out) factors += t.size();
            if (factors > 4 * GPX_MAX_PARTS) {
                gpx_set_error("kspec: the expanded sum of products has more than %d factors",
                              GPX_MAX_PARTS);
                return -1;
            }
        }
        return 0;
    }
    KPart leaf;
    GPX_TRY(fill_leaf(k, d, *nhyper, &leaf));
    *nhyper += leaf.nhyper;
    leaves.push_back(leaf);
    out->push_back({(int)leaves.size() - 1});
    return 0;
}

int gpx_flatten_kspec(const gpx_kspec *k, int64_t d, KParams *out)
{
    if (!k) {
        gpx_set_error("kspec: null");
        return -1;
    }
    if (d < 1 || d > GPX_MAX_DIM) {
        gpx_set_error("kspec: ndim %lld outside 1..%d", (long long)d, GPX_MAX_DIM);
        return -1;
    }
    out->nparts = 0;
    out->nhyper = 0;
    out->ndim = (int)d;
    out->nprod = 0;
    std::vector<KPart> leaves;
    Terms terms;
    int nhyper = 0;
    GPX_TRY(expand(k, d, leaves, &nhyper, &terms));
    out->nhyper = nhyper;
    std::vector<char> seen(leaves.size(), 0);
    int group = 0;
    for (const auto &t : terms) {
        for (int leaf : t) {
            if (out->nparts >= GPX_MAX_PARTS) {
                gpx_set_error("kspec: the expanded sum of products has more than %d factors",
                              GPX_MAX_PARTS);
                return -1;
            }
            KPart &p = out->part[out->nparts++];
            p = leaves[leaf];
            p.group = group;
            p.dup = seen[leaf];
            seen[leaf] = 1;
        }
        if (t.size() > 1) out->nprod += (int)t.size();
        ++group;
    }
    return 0;
}

// Rebuild a kspec tree that points into a flat hyper vector (batched thetas).
static int rebind(const gpx_kspec *k, const double *&hyper,
                  std::vector<gpx_kspec> &store, size_t &cursor, gpx_kspec *out)
{
    *out = *k;
    if (k->kind == GPX_SUM || k->kind == GPX_PRODUCT) {
        size_t base = cursor;
        cursor += k->nparts;
        for (int i = 0; i < k->nparts; ++i)
            GPX_TRY(rebind(&k->parts[i], hyper, store, cursor, &store[base + i]));
        out->parts = &store[base];
        out->hyper = nullptr;
    } else {
        out->hyper = hyper;
        hyper += k->nhyper;
    }
    return 0;
}

static size_t count_nodes(const gpx_kspec *k)
{
    size_t n = 0;
    if (k->kind == GPX_SUM || k->kind == GPX_PRODUCT)
        for (int i = 0; i < k->nparts; ++i) n += 1 + count_nodes(&k->parts[i]);
    return n;
}

int gpx_kspec_with_hyper(const gpx_kspec *k, const double *hyper,
                         std::vector<gpx_kspec> &store, gpx_kspec *out)
{
    store.assign(count_nodes(k) + 1, gpx_kspec{});
    size_t cursor = 0;
    return rebind(k, hyper, store, cursor, out);
}

// exp() of the trace-gradient kernels: the argument is 2 log sf - D2/2 or 2 log sf - r,
// never large and positive, so there are no special cases --
// x is clamped at -1000, where v_ldexp_f64 underflows to 0 by itself, overflow ends in inf
// through the same ldexp, NaN stays NaN (the clamp is a compare + select, not v_max_f64,
// which would turn a NaN input into exp(-1000) = 0). The library exp spent 9 v_mov_b64
// per call copying coefficients (the compiler lowers fma(r, p, c) with a constant addend to
// a copy + v_fmac, whose addend is its destination) and 6 instructions on overflow /
// underflow selects: 70 -> 56 VALU instructions per SE pair of the trace kernel.
// Polynomial: exp(r) = 1 + r + r^2 q(r), |r| <= ln2/2, q of degree 9 (tools/exp_coeffs.py:
// relative error 1.6e-17 before rounding).
__device__ __forceinline__ double gpx_fma3(double a, double b, double c)
{
    double d;                                            // VOP3 form: d need not be c
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double gpx_exp(double x)
{
    x = x < -1000.0 ? -1000.0 : x;
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);   // ln2 = hi + lo, k * hi exact
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 0x1.af38a9b0ec855p-26;
    p = gpx_fma3(r, p, 0x1.289185613a3d6p-22);
    p = gpx_fma3(r, p, 0x1.71de0dae63bb3p-19);
    p = gpx_fma3(r, p, 0x1.a019b90d2ae7ap-16);
    p = gpx_fma3(r, p, 0x1.a01a01a7c41d5p-13);
    p = gpx_fma3(r, p, 0x1.6c16c1788bd90p-10);
    p = gpx_fma3(r, p, 0x1.11111111109b3p-7);
    p = gpx_fma3(r, p, 0x1.5555555553d63p-5);
    p = gpx_fma3(r, p, 0x1.5555555555556p-3);
    p = gpx_fma3(r, p, 0x1.0000000000001p-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)k);
}

// ---- device: one primitive part on one pair ---------------------------------
template <typename T> struct Math;
// fp64: correctly-rounded-class ocml functions and true divisions, so that values
// match the NumPy reference to ~1 ulp. (gpx_exp above in the build: measured, no gain --
// 11.8k instead of 11.5k instructions in the kernel, N = 32768 SE build 2.09 against
// 1.84 ms -- the library exp does not pay for coefficient copies here.) fp32 (BASELINE config 5, an HBM-write
// bound build at rel 1e-5 / abs 1e-6): hardware v_exp_f32 / v_sin_f32 forms and a
// reciprocal multiply, ~35 instructions per pair instead of ~110.
template <> struct Math<double> {
    static __device__ __forceinline__ double over(double a, double b) { return a / b; }
    static __device__ __forceinline__ double exp_(double x) { return exp(x); }
    static __device__ __forceinline__ double pow_(double x, double y) { return pow(x, y); }
    static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
    static __device__ __forceinline__ double sin_(double x) { return sin(x); }
    static __device__ __forceinline__ double cos_(double x) { return cos(x); }
};
template <> struct Math<float> {
    static __device__ __forceinline__ float over(float a, float b)
    {
        return a * __builtin_amdgcn_rcpf(b);
    }
    static __device__ __forceinline__ float exp_(float x) { return __expf(x); }
    static __device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
    static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float sin_(float x) { return __sinf(x); }
    static __device__ __forceinline__ float cos_(float x) { return __cosf(x); }
};

// value of one part given its (scaled) squared distance D2
template <typename T>
__device__ __forceinline__ T part_value(int kind, T two_logsf, T sf2, T ell,
                                        T period, T alpha, T D2)
{
    typedef Math<T> M;
    switch (kind) {
    case GPX_RQ:                                   // rq.py:54-61
        return sf2 * M::pow_(1 + D2 / (2 * alpha), -alpha);
    case GPX_SE:                                   // se.py:55
        return M::exp_(two_logsf - D2 / 2);
    case GPX_MATERN1: {                            // matern.py:71-74, _f :44-48
        T r = M::sqrt_(D2);
        return M::exp_(two_logsf - r);
    }
    case GPX_MATERN3: {
        T r = M::sqrt_(D2);
        return M::exp_(two_logsf - r) * (1 + r);
    }
    case GPX_MATERN5: {
        T r = M::sqrt_(D2);
        return M::exp_(two_logsf - r) * (1 + r * (1 + r / T(3)));
    }
    default: {                                     // periodic.py:57-58, in its order:
        T u = M::sqrt_(D2) * T(M_PI) / period;      // sqrt(D) * pi / p
        T s = M::over(M::sin_(u), ell);
        return sf2 * M::exp_(-2 * (s * s));
    }
    }
}

// ---- kbuild ------------------------------------------------------------------
template <typename T> struct Vec4;
template <> struct Vec4<double> { typedef double4 type; };
template <> struct Vec4<float> { typedef float4 type; };

// One part on a 4 x 4 register tile of (scaled) squared distances. The kernel family
// is a template argument: the switch over part.kind runs once per tile, not once
// per pair, and everything that only depends on the hyperparameters is folded into
// constants in front of the 16-pair loop.
//   fp64: the expressions of the reference, ocml exp / sqrt / sin (~1 ulp).
//   fp32 (BASELINE config 5, tolerance rel 1e-5 / abs 1e-6): one v_exp_f32 per SE
//        pair (exp2 of one FMA), v_sqrt + v_sin + v_exp per periodic pair.
template <typename T, int KIND> struct PartTile;

template <int KIND> struct PartTile<double, KIND> {
    static __device__ __forceinline__ void eval(const KPart &part, const double (&D2)[4][4],
                                                double (&v)[4][4])
    {
        const double tl = part.two_logsf, sf2 = part.sf2, ell = part.ell,
                     pp = part.period, al = part.alpha;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                v[a][b] = part_value<double>(KIND, tl, sf2, ell, pp, al, D2[a][b]);
    }
};

#define GPX_LOG2E_F 1.44269504088896340736f
typedef float f32x2 __attribute__((ext_vector_type(2)));   // v_pk_*_f32 operands
template <int KIND> struct PartTile<float, KIND> {
    static __device__ __forceinline__ void eval(const KPart &part, const float (&D2)[4][4],
                                                float (&v)[4][4])
    {
        const float c0 = (float)(part.two_logsf * 1.44269504088896340736);   // log2 sf^2
        if (KIND == GPX_SE) {                              // exp(2 log sf - D2 / 2)
            const f32x2 c1 = {-0.5f * GPX_LOG2E_F, -0.5f * GPX_LOG2E_F}, c0v = {c0, c0};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; b += 2) {
                    const f32x2 d = {D2[a][b], D2[a][b + 1]};
                    const f32x2 e = __builtin_elementwise_fma(d, c1, c0v);
                    v[a][b] = __builtin_amdgcn_exp2f(e.x);
                    v[a][b + 1] = __builtin_amdgcn_exp2f(e.y);
                }
        } else if (KIND == GPX_PERIODIC) {
            // sf^2 exp(-2 sin^2(pi r / p) / ell^2); v_sin_f32 takes revolutions:
            // pi r / p = 2 pi (r / (2 p))
            const float rev1 = (float)(part.pi_over_p * (0.5 / M_PI));
            const float c11 = (float)(-2.0 * 1.44269504088896340736 / (part.ell * part.ell));
            const f32x2 rev = {rev1, rev1}, c1 = {c11, c11}, c0v = {c0, c0};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; b += 2) {
                    f32x2 r = {__builtin_amdgcn_sqrtf(D2[a][b]), __builtin_amdgcn_sqrtf(D2[a][b + 1])};
                    r = r * rev;
                    f32x2 sn = {__builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r.x)),
                                __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r.y))};
                    const f32x2 e = __builtin_elementwise_fma(sn * sn, c1, c0v);
                    v[a][b] = __builtin_amdgcn_exp2f(e.x);
                    v[a][b + 1] = __builtin_amdgcn_exp2f(e.y);
                }
        } else if (KIND == GPX_RQ) {                       // sf^2 (1 + D2 / 2a)^-a
            const float ia = (float)(0.5 / part.alpha), na = (float)(-part.alpha);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    v[a][b] = __builtin_amdgcn_exp2f(__builtin_fmaf(
                        na, __builtin_amdgcn_logf(__builtin_fmaf(D2[a][b], ia, 1.0f)), c0));
        } else {                                           // Matern: sf^2 e^-r f(r)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float r = __builtin_amdgcn_sqrtf(D2[a][b]);
                    const float S = __builtin_amdgcn_exp2f(__builtin_fmaf(r, -GPX_LOG2E_F, c0));
                    const float f = KIND == GPX_MATERN1 ? 1.0f
                                    : (KIND == GPX_MATERN3 ? 1.0f + r
                                                           : 1.0f + r * (1.0f + r * (1.0f / 3)));
                    v[a][b] = S * f;
                }
        }
    }
};

template <typename T>
__device__ __forceinline__ void part_tile(const KPart &part, const T (&D2)[4][4], T (&v)[4][4])
{
    switch (part.kind) {
    case GPX_SE: PartTile<T, GPX_SE>::eval(part, D2, v); break;
    case GPX_MATERN1: PartTile<T, GPX_MATERN1>::eval(part, D2, v); break;
    case GPX_MATERN3: PartTile<T, GPX_MATERN3>::eval(part, D2, v); break;
    case GPX_MATERN5: PartTile<T, GPX_MATERN5>::eval(part, D2, v); break;
    case GPX_RQ: PartTile<T, GPX_RQ>::eval(part, D2, v); break;
    default: PartTile<T, GPX_PERIODIC>::eval(part, D2, v); break;
    }
}

// D2[a][b] += (xi[a] - xj[b])^2 for a 4 x 4 block (_distances.py:41, direct form). fp32
// works on pairs of columns: v_pk_add_f32 / v_pk_fma_f32 do two lanes' worth per issue
// slot (the fp32 vector peak of the part is quoted for packed math), which halves the
// instruction count of the part of the config-5 build that is not transcendental.
__device__ __forceinline__ void d2_accum(const double (&xi)[4], const double (&xj)[4],
                                         double (&D2)[4][4])
{
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double df = xi[a] - xj[b];
            D2[a][b] += df * df;
        }
}
__device__ __forceinline__ void d2_accum(const float (&xi)[4], const float (&xj)[4],
                                         float (&D2)[4][4])
{
    const f32x2 x01 = {xj[0], xj[1]}, x23 = {xj[2], xj[3]};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x2 xa = {xi[a], xi[a]};
        const f32x2 d01 = xa - x01, d23 = xa - x23;
        f32x2 s01 = {D2[a][0], D2[a][1]}, s23 = {D2[a][2], D2[a][3]};
        s01 = __builtin_elementwise_fma(d01, d01, s01);
        s23 = __builtin_elementwise_fma(d23, d23, s23);
        D2[a][0] = s01.x; D2[a][1] = s01.y; D2[a][2] = s23.x; D2[a][3] = s23.y;
    }
}

// One workgroup = W side-by-side 64 x 64 tiles of one tile row (round 3). With one tile
// per workgroup the config-5 build is 131 000 workgroups of 32 KB of stores each, and a
// pure store kernel of that shape runs at the same 1.0-1.2 ms whatever arithmetic it
// does: the workgroup launch rate (~110 per microsecond on this part) sets the pace,
// not HBM (tools/probe_dispatch.hip: 1 / 2 / 4 tiles per workgroup 3.6 / 4.7 / 5.3 TB/s).
// All inputs of the run are staged up front (one barrier), then the W tiles are
// evaluated and stored one after the other without further global loads or barriers
// (a strip loop that staged per tile serialised load -> barrier -> compute -> store and
// was slower than one tile per workgroup).
// MB: member-batched launch (round 4). blockIdx.z is the member: its hyperparameters and
// noise variance come from its record in device memory (mp), its output matrices lie
// mstride elements behind those of member 0; the inputs are shared.
template <typename T, int W, bool MB>
__global__ __launch_bounds__(256) void kbuild_kernel(
    KParams kp_arg, const T *__restrict__ X1, int n1, const T *__restrict__ X2, int n2,
    int d, T *__restrict__ out, long long ldo, int sym, int upper_only, T diag_add,
    int joff, T *__restrict__ out_off, int mirror, int itile0, int ntj, int rows, int tri,
    long long koff, const MemberParams *__restrict__ mp, long long mstride)
{
    const KParams &kp = MB ? mp[blockIdx.z].kp : kp_arg;
    if (MB) {
        diag_add = (T)mp[blockIdx.z].sn2;
        out += (long long)blockIdx.z * mstride;
        if (out_off) out_off += (long long)blockIdx.z * mstride;
    }
    // joff: global index of column 0 (a column strip of a symmetric matrix; a multiple
    // of 128); itile0: first tile row of this launch (a row strip); ntj: 64-column
    // tiles; rows: input rows staged per block (all parts at once when they fit, else d);
    // tri: 1-D grid over the live (tile row, tile group) pairs of a triangular build,
    // koff: pairs of the tile rows above this launch's first one (row strips)
    extern __shared__ __attribute__((aligned(32))) char kb_smem[];
    constexpr int XW = KT * W;                             // columns of a run
    T *xi_base = reinterpret_cast<T *>(kb_smem);          // [rows][KT]
    T *xj_base = xi_base + rows * KT;                      // [rows][XW]

    const int G = (ntj + W - 1) / W;                       // tile groups per row
    int bi, g;
    if (tri) {
        // Rows come in bands of W with G - q live groups each (band q = row / W): both
        // the mirror rule (tile column >= tile row) and the upper_only rule of the
        // 128-tile engine (W even) keep exactly the groups g >= q.
        const long long k = blockIdx.x + koff;
        const float s2 = 2.0f * G + 1.0f;
        int q = (int)((s2 - sqrtf(fmaxf(s2 * s2 - 8.0f * (float)k / W, 0.0f))) * 0.5f);
        q = max(0, min(q, G - 1));
        while ((long long)W * ((long long)G * q - (long long)q * (q - 1) / 2) > k) --q;
        while ((long long)W * ((long long)G * (q + 1) - (long long)(q + 1) * q / 2) <= k) ++q;
        const int rem = (int)(k - (long long)W * ((long long)G * q - (long long)q * (q - 1) / 2));
        const int per = G - q;
        bi = q * W + rem / per;
        g = q + rem % per;
    } else {
        bi = blockIdx.y + itile0;
        g = blockIdx.x;
        // dead groups of a triangular build on a rectangular grid (strips)
        const int last = min(ntj, (g + 1) * W) - 1 + joff / KT;
        if (mirror ? last < bi : (upper_only && (last >> 1) < (bi >> 1))) return;
    }
    // Thread (tx, ty) owns the 4 x 4 block at rows 4 ty, columns 4 tx of every tile. A
    // wave is an 8 x 8 patch of threads (the four waves tile the 16 x 16 thread grid
    // 2 x 2): a row store of the wave is 8 lanes x 16 B = 128 contiguous bytes, and so is
    // a store of the TRANSPOSED block, which a thread forms in its own registers
    // (column b of its block is 4 consecutive elements of row 4 tx + b of the mirror
    // image) -- the mirror mode needs no trip through LDS and no barrier (round 2
    // staged the transposed tile in LDS: two barriers per tile and 1.3e8 bank conflicts
    // per config-5 build).
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tx = (wv & 1) * 8 + (lane & 7), ty = (wv >> 1) * 8 + (lane >> 3);
    const int i0 = bi * KT, jg0 = g * XW;
    const int nw = min(W, ntj - g * W);                    // tiles of this run
    const bool products = kp.nprod != 0;
    const bool together = kp.nparts * d <= GPX_MAX_DIM;    // else W == 1 (host)

    // input rows divided by a part's lengthscales (_distances.py:17-23), k-major
    auto stage = [&](int p, int rowoff) {
        for (int r = tid; r < KT + nw * KT; r += 256) {
            const KPart &part = kp.part[p];
            const bool second = r >= KT;
            const int rr = second ? r - KT : r;
            const T *src = second ? X2 + (size_t)min(jg0 + rr, n2 - 1) * d
                                  : X1 + (size_t)min(i0 + rr, n1 - 1) * d;
            T *dst = second ? xj_base + rr : xi_base + rr;
            const int stride = second ? XW : KT;
            for (int c = 0; c < d; ++c) dst[(rowoff + c) * stride] = src[c] / (T)part.scale[c];
        }
    };
    if (together) {
        for (int p = 0; p < kp.nparts; ++p) stage(p, p * d);
        __syncthreads();
    }
    for (int w = 0; w < nw; ++w) {
        const int bj = g * W + w;
        const int j0 = bj * KT;
        const int bjg = bj + joff / KT;                    // global column tile
        if (mirror ? bj < bi : (upper_only && (bjg >> 1) < (bi >> 1))) continue;
        const T *xj_cur = xj_base + w * KT;
        // K = sum over groups of the product of the group's parts (_combo.py:103-131);
        // plain sums (no products anywhere) add straight into acc
        T acc[4][4], prod[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                acc[a][b] = 0;
                prod[a][b] = 0;
            }
        for (int p = 0; p < kp.nparts; ++p) {
            const KPart &part = kp.part[p];
            const bool opens = p == 0 || part.group != kp.part[p - 1].group;
            const int c0 = together ? p * d : 0;
            if (!together) {                               // part by part through one stage
                __syncthreads();
                stage(p, 0);
                __syncthreads();
            }
            T D2[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) D2[a][b] = 0;
#pragma unroll 4
            for (int c = c0; c < c0 + d; ++c) {
                T xi[4], xj[4];
                const typename Vec4<T>::type u =
                    *reinterpret_cast<const typename Vec4<T>::type *>(&xi_base[c * KT + 4 * ty]);
                xi[0] = u.x; xi[1] = u.y; xi[2] = u.z; xi[3] = u.w;
                const typename Vec4<T>::type v =
                    *reinterpret_cast<const typename Vec4<T>::type *>(&xj_cur[c * XW + 4 * tx]);
                xj[0] = v.x; xj[1] = v.y; xj[2] = v.z; xj[3] = v.w;
                d2_accum(xi, xj, D2);
            }
            T val[4][4];
            part_tile<T>(part, D2, val);
            if (!products) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] += val[a][b];
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (opens) {
                            acc[a][b] += prod[a][b];
                            prod[a][b] = val[a][b];
                        } else {
                            prod[a][b] *= val[a][b];
                        }
                    }
            }
        }
        if (products) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += prod[a][b];
        }

        T *dst = (out_off && (bi >> 1) != (bjg >> 1)) ? out_off : out;
        // interior tiles (no padding, not on the diagonal of a symmetric build) keep
        // their values as they are
        const bool edge = i0 + KT > n1 || j0 + KT > n2 || (sym && bi == bjg);
        if (edge) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int gi = i0 + 4 * ty + a;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int gj = j0 + 4 * tx + b;
                    T x = acc[a][b];
                    if (sym) {
                        if (gi == gj + joff) x += diag_add;          // exact.py:52
                        if (gi >= n1 || gj >= n2) x = (gi == gj + joff) ? T(1) : T(0);
                    } else if (gi >= n1 || gj >= n2) {
                        x = 0;
                    }
                    acc[a][b] = x;
                }
            }
        }
        T *drow = dst + (size_t)(i0 + 4 * ty) * ldo + j0 + 4 * tx;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            typename Vec4<T>::type o;
            o.x = acc[a][0]; o.y = acc[a][1]; o.z = acc[a][2]; o.w = acc[a][3];
            *reinterpret_cast<typename Vec4<T>::type *>(drow + (size_t)a * ldo) = o;
        }
        if (mirror && bj != bi) {
            T *trow = out + (size_t)(j0 + 4 * tx) * ldo + i0 + 4 * ty;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                typename Vec4<T>::type o;
                o.x = acc[0][b]; o.y = acc[1][b]; o.z = acc[2][b]; o.w = acc[3][b];
                *reinterpret_cast<typename Vec4<T>::type *>(trow + (size_t)b * ldo) = o;
            }
        }
    }
}

static size_t kbuild_lds(size_t elem, int rows, int W, bool)
{
    return (size_t)rows * KT * (1 + W) * elem;
}

template <typename T, int W> static int kbuild_attr()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&kbuild_kernel<T, W, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&kbuild_kernel<T, W, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    return 0;
}

int gpx_kmat_init()
{
    GPX_TRY((kbuild_attr<double, 1>()));
    GPX_TRY((kbuild_attr<double, 2>()));
    GPX_TRY((kbuild_attr<double, 4>()));
    GPX_TRY((kbuild_attr<double, 8>()));
    GPX_TRY((kbuild_attr<float, 8>()));
    GPX_TRY((kbuild_attr<float, 1>()));
    GPX_TRY((kbuild_attr<float, 2>()));
    GPX_TRY((kbuild_attr<float, 4>()));
    return 0;
}

// launch with W tiles per workgroup chosen by the stage size (GPX_KBUILD_W forces 1 / 2 / 4)
template <typename T>
static int kbuild_launch(hipStream_t s, const KParams &kp, const T *X1, int n1, const T *X2,
                         int n2, int d, T *out, long long ldo, int sym, int upper_only,
                         double diag_add, int joff, T *out_off, int mirror, int itile0,
                         int ntj, int tile_rows, bool triangular,
                         const MemberBatch *mb = nullptr)
{
    static const int forced = getenv("GPX_KBUILD_W") ? atoi(getenv("GPX_KBUILD_W")) : 0;
    const bool together = kp.nparts * d <= GPX_MAX_DIM;
    const int rows = together ? kp.nparts * d : d;
    int W = together ? 4 : 1;
    if (together && (forced == 1 || forced == 2 || forced == 4 || forced == 8)) W = forced;
    while (W > 1 && (kbuild_lds(sizeof(T), rows, W, mirror != 0) > 64 * 1024 || ntj < W)) W /= 2;
    // the 1-D triangular grid needs an even W (see the kernel) and whole bands of W rows
    const int tri = triangular && W > 1 && joff == 0 && itile0 % W == 0;
    const int G = (ntj + W - 1) / W;
    const int members = mb ? mb->count : 1;
    const MemberParams *mp = mb ? mb->params : nullptr;
    const long long mstride = mb ? mb->mstride : 0;
    dim3 grid(G, tile_rows, members);
    long long koff = 0;
    if (tri) {
        long long live = 0;
        for (int r = 0; r < itile0; ++r) koff += G - r / W;
        for (int r = itile0; r < itile0 + tile_rows; ++r) live += G - r / W;
        if (live <= 0) return 0;
        grid = dim3((unsigned)live, 1, members);
    }
    const size_t lds = kbuild_lds(sizeof(T), rows, W, mirror != 0);
#define GPX_KB_LAUNCH(WW)                                                                  \
    do {                                                                                   \
        if (mp)                                                                            \
            hipLaunchKernelGGL((kbuild_kernel<T, WW, true>), grid, dim3(256), lds, s, kp, X1, n1, \
                               X2, n2, d, out, ldo, sym, upper_only, (T)diag_add, joff, out_off, \
                               mirror, itile0, ntj, rows, tri, koff, mp, mstride);         \
        else                                                                               \
            hipLaunchKernelGGL((kbuild_kernel<T, WW, false>), grid, dim3(256), lds, s, kp, X1, n1, \
                               X2, n2, d, out, ldo, sym, upper_only, (T)diag_add, joff, out_off, \
                               mirror, itile0, ntj, rows, tri, koff, mp, mstride);         \
    } while (0)
    if (W == 8) GPX_KB_LAUNCH(8);
    else if (W == 4) GPX_KB_LAUNCH(4);
    else if (W == 2) GPX_KB_LAUNCH(2);
    else GPX_KB_LAUNCH(1);
#undef GPX_KB_LAUNCH
    GPX_HIP(hipGetLastError());
    return 0;
}

template <typename T>
int gpx_kbuild(hipStream_t s, const KParams &kp, const T *X1, int n1, int np1,
               const T *X2, int n2, int np2, int d, T *out, long long ldo,
               bool sym, bool upper_only, double diag_add, T *out_offdiag, int row0, int rows,
               const MemberBatch *mb)
{
    if (rows < 0) rows = np1 - row0;
    if (np1 % KT || np2 % KT || n1 < 1 || n2 < 1 || row0 < 0 || row0 % KT || rows % KT ||
        row0 + rows > np1) {
        gpx_set_error("kbuild: bad shape n1=%d np1=%d n2=%d np2=%d rows [%d, +%d)", n1, np1,
                      n2, np2, row0, rows);
        return -1;
    }
    if (rows == 0) return 0;
    GPX_TRY(gpx_test_jitter(s));
    const bool strip = row0 != 0 || rows != np1;
    // the full square K(X, X): upper tiles only, each stored twice
    const int mirror = !strip && !sym && !upper_only && !out_offdiag && X1 == X2 && n1 == n2 &&
                       np1 == np2;
    const bool triangular = (mirror || (sym && upper_only)) && np1 == np2;
    return kbuild_launch<T>(s, kp, X1, n1, X2, n2, d, out, ldo, sym ? 1 : 0, upper_only ? 1 : 0,
                            diag_add, 0, out_offdiag, mirror, row0 / KT, np2 / KT, rows / KT,
                            triangular, mb);
}

// columns [j0, j0 + npc) of the symmetric n x n matrix K + diag_add I (identity in
// the padding), all np rows: the strip an appended observation touches
int gpx_kbuild_strip(hipStream_t s, const KParams &kp, const double *X, int n, int np,
                     int j0, int npc, int d, double *out, long long ldo, double diag_add,
                     double *out_offdiag)
{
    if (np % KT || npc % KT || j0 < 0 || j0 >= n || j0 % GPX_TILE) {
        gpx_set_error("kbuild_strip: bad shape n=%d np=%d j0=%d npc=%d", n, np, j0, npc);
        return -1;
    }
    return kbuild_launch<double>(s, kp, X, n, X + (size_t)j0 * d, n - j0, d, out + j0, ldo, 1, 0,
                                 diag_add, j0, out_offdiag ? out_offdiag + j0 : (double *)nullptr,
                                 0, 0, npc / KT, np / KT, false);
}
template int gpx_kbuild<double>(hipStream_t, const KParams &, const double *, int, int,
                                const double *, int, int, int, double *, long long,
                                bool, bool, double, double *, int, int, const MemberBatch *);
template int gpx_kbuild<float>(hipStream_t, const KParams &, const float *, int, int,
                               const float *, int, int, int, float *, long long, bool,
                               bool, double, float *, int, int, const MemberBatch *);

// ---- gradient pieces shared by kgrad and trace_grad ---------------------------
// For SE / Matern parts: K, and M such that dK/dlog ell_c = M * dd_c / r_div with
// the reference's guard (matern.py:87-90); iso: isoval (se.py:62, matern.py:85).
struct RadialGrad {
    double K;        // kernel value
    double Mv;       // SE: K ; Matern: S * _df(r)
    double rdiv;     // SE: 1 ; Matern: r
    double isoval;   // SE: K * D2 ; Matern: Mv * r
    bool zero;       // Matern: r < 1e-12  -> ARD slices are 0
    double xval;     // RQ: dK / dlog alpha (rq.py:84)
};
template <bool WITH_RQ>
__device__ __forceinline__ RadialGrad radial_grad_t(int kind, double two_logsf, double sf2,
                                                    double alpha, double D2)
{
    RadialGrad g;
    g.xval = 0.0;
    if (WITH_RQ && kind == GPX_RQ) {                   // rq.py:63-84
        const double E = 1 + 0.5 * D2 / alpha;
        g.K = sf2 * pow(E, -alpha);
        g.Mv = g.K;
        g.rdiv = E;
        g.isoval = g.K * D2 / E;
        g.zero = false;
        g.xval = 0.5 * g.isoval - alpha * g.K * log(E);
    } else if (kind == GPX_SE) {                              // se.py:57-66
        g.K = exp(two_logsf - D2 / 2);
        g.Mv = g.K;
        g.rdiv = 1.0;
        g.isoval = g.K * D2;
        g.zero = false;
    } else {                                           // matern.py:76-90
        const double r = sqrt(D2);
        const double S = exp(two_logsf - r);
        const double f = kind == GPX_MATERN1 ? 1.0
                         : (kind == GPX_MATERN3 ? 1 + r : 1 + r * (1 + r / 3.));
        const double df = kind == GPX_MATERN1 ? 1.0
                          : (kind == GPX_MATERN3 ? r : r * (1 + r) / 3.);
        g.K = S * f;
        g.Mv = S * df;
        g.rdiv = r;
        g.isoval = g.Mv * r;
        g.zero = r < 1e-12;
    }
    return g;
}
__device__ __forceinline__ RadialGrad radial_grad(int kind, double two_logsf, double sf2,
                                                  double alpha, double D2)
{
    return radial_grad_t<true>(kind, two_logsf, sf2, alpha, D2);
}
struct PeriodicGrad { double g0, g1, g2; };
__device__ __forceinline__ PeriodicGrad periodic_grad(double sf2, double ell,
                                                      double period, double D2)
{                                                      // periodic.py:61-74
    const double u = sqrt(D2) * M_PI / period;
    const double R = sin(u) / ell;
    const double S = R * R;
    const double E = 2 * sf2 * exp(-2 * S);
    PeriodicGrad g;
    g.g0 = E;
    g.g1 = 2 * E * S;
    g.g2 = 2 * E * R * u * cos(u) / ell;
    return g;
}

// ---- products: the factor in front of a part's derivatives ---------------------
// d/dh (prod_q k_q) = (prod_{q != p} k_q) dk_p/dh (_combo.py:133-146): value of
// the other parts of p's group on the pair (x1, x2); 1 for plain sums.
__device__ __forceinline__ double part_value_pair(const KPart &q, const double *x1,
                                                  const double *x2, int d)
{
    double D2 = 0.0;
    for (int c = 0; c < d; ++c) {
        const double df = x1[c] / q.scale[c] - x2[c] / q.scale[c];
        D2 += df * df;
    }
    return part_value<double>(q.kind, q.two_logsf, q.sf2, q.ell, q.period, q.alpha, D2);
}

__device__ __forceinline__ double group_factor(const KParams &kp, int p, const double *x1,
                                               const double *x2, int d)
{
    double f = 1.0;
    if (kp.nprod == 0) return f;
    for (int q = 0; q < kp.nparts; ++q)
        if (q != p && kp.part[q].group == kp.part[p].group)
            f *= part_value_pair(kp.part[q], x1, x2, d);
    return f;
}

// ---- kgrad (API path, one thread per pair) ----------------------------------
__global__ __launch_bounds__(256) void kgrad_kernel(
    KParams kp, const double *__restrict__ X1, int n1, const double *__restrict__ X2,
    int n2, int d, double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n2) return;
    const size_t plane = (size_t)n1 * n2;
    double *o = out + (size_t)i * n2 + j;
    const double *xi = X1 + (size_t)i * d, *xj = X2 + (size_t)j * d;
    for (int p = 0; p < kp.nparts; ++p) {
        const KPart &part = kp.part[p];
        double D2 = 0;
        for (int c = 0; c < d; ++c) {
            const double df = xi[c] / part.scale[c] - xj[c] / part.scale[c];
            D2 += df * df;
        }
        double *oh = o + (size_t)part.hoff * plane;
        const double fac = group_factor(kp, p, xi, xj, d);
        // a primitive repeated by the expansion of a sum inside a product: its
        // contributions add up in the same slots
#define GPX_PUT(slot, val)                                                       \
    do {                                                                         \
        double *q_ = oh + (size_t)(slot) * plane;                                \
        *q_ = (part.dup ? *q_ : 0.0) + (val);                                    \
    } while (0)
        if (part.kind == GPX_PERIODIC) {
            const PeriodicGrad g = periodic_grad(part.sf2, part.ell, part.period, D2);
            GPX_PUT(0, fac * g.g0);
            GPX_PUT(1, fac * g.g1);
            GPX_PUT(2, fac * g.g2);
            continue;
        }
        const RadialGrad g = radial_grad(part.kind, part.two_logsf, part.sf2, part.alpha, D2);
        GPX_PUT(0, fac * (2 * g.K));
        if (part.iso) {
            GPX_PUT(1, fac * g.isoval);
        } else {
            for (int c = 0; c < d; ++c) {
                const double df = xi[c] / part.scale[c] - xj[c] / part.scale[c];
                GPX_PUT(1 + c, g.zero ? 0.0 : fac * ((g.Mv * (df * df)) / g.rdiv));
            }
        }
        if (part.kind == GPX_RQ) GPX_PUT(part.nhyper - 1, fac * g.xval);
#undef GPX_PUT
    }
}

int gpx_kgrad(hipStream_t s, const KParams &kp, const double *X1, int n1,
              const double *X2, int n2, int d, double *out)
{
    dim3 grid((n2 + 255) / 256, n1);
    hipLaunchKernelGGL(kgrad_kernel, grid, dim3(256), 0, s, kp, X1, n1, X2, n2, d, out);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- trace_grad --------------------------------------------------------------
// Upper-triangle tiles (KT x KT) of Kinv. Thread (lane j = tid & 63, row group
// ig = tid >> 6) owns column j0 + j and rows i0 + ig*16 .. +15. x_j (scaled)
// lives in registers, x_i is broadcast from LDS. Accumulators per part:
// a_sf (sum w q K), a_e[c] (sum w q dK/dlog ell_c); tr(Q) once.
#define TG_MAXACC (GPX_MAX_HYPER + 1)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// MODE 1: every part is SE or Matern and nothing multiplies (the configurations of
// BASELINE.json): the RQ / periodic / product code paths are compiled out of the
// pair loop (1.5 vs 2.2 ms at N = 16384, D = 8 with them in)
template <int DMAX, int MODE, bool MB>
__global__ __launch_bounds__(256) void trace_grad_kernel(
    KParams kp_arg, const double *__restrict__ X, int n, int d,
    const double *__restrict__ Kinv, int ld, const double *__restrict__ alpha,
    double *__restrict__ partial, int nacc, const MemberParams *__restrict__ mp,
    long long mstride, long long vstride, long long pstride)
{
    const KParams &kp = MB ? mp[blockIdx.z].kp : kp_arg;
    if (MB) {
        Kinv += (long long)blockIdx.z * mstride;
        alpha += (long long)blockIdx.z * vstride;
        partial += (long long)blockIdx.z * pstride;
    }
    // linear block id -> upper-triangular tile (bi <= bj)
    const int T = gridDim.y;                    // tiles per side
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int blin = bi * T + bj;
    double *pout = partial + (size_t)blin * nacc;
    const int tid = threadIdx.x;
    if (bj < bi) {
        for (int h = tid; h < nacc; h += 256) pout[h] = 0.0;
        return;
    }
    __shared__ double xi_s[KT][DMAX + 1];
    __shared__ double red[4][DMAX + 3];

    const int lane = tid & 63, ig = tid >> 6;
    const int i0 = bi * KT, j0 = bj * KT;
    const int gj = j0 + lane;
    const int cj = min(gj, n - 1);
    const double aj = alpha[cj];

    // q weights for this thread's 16 pairs (shared by all parts)
    double wq[16];
    double trq = 0.0;
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
        const int gi = i0 + ig * 16 + ii;
        double w = 0.0;
        if (gi < n && gj < n) w = gi < gj ? 2.0 : (gi == gj ? 1.0 : 0.0);
        double q = 0.0;
        if (w != 0.0) {
            q = Kinv[(size_t)gi * ld + gj] - alpha[gi] * aj;   // exact.py:129-130
            if (gi == gj) trq += q;
        }
        wq[ii] = w * q;
    }

    for (int p = 0; p < kp.nparts; ++p) {
        const KPart &part = kp.part[p];
        __syncthreads();
        // dimensions beyond d are staged as zeros on both sides: the pair loop below
        // runs over all DMAX of them without a branch (a runtime `c < d` per
        // dimension serialises the LDS reads: 2.2 -> see DESIGN.md)
        for (int e = tid; e < KT * DMAX; e += 256) {
            const int r = e / DMAX, c = e - r * DMAX;
            const int gi = min(i0 + r, n - 1);
            xi_s[r][c] = c < d ? X[(size_t)gi * d + c] / part.scale[c] : 0.0;
        }
        double xj[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            xj[c] = c < d ? X[(size_t)cj * d + c] / part.scale[c] : 0.0;
        __syncthreads();

        double a_sf = 0.0, a_x = 0.0, a_e[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) a_e[c] = 0.0;

        for (int ii = 0; ii < 16; ++ii) {
            double t = wq[ii];
            // product kernels: the other factors of this part's group, evaluated
            // from the unscaled inputs (rare path, straight from L2)
            if (MODE == 0 && kp.nprod != 0 && t != 0.0)
                t *= group_factor(kp, p, X + (size_t)min(i0 + ig * 16 + ii, n - 1) * d,
                                  X + (size_t)cj * d, d);
            const double *xi = xi_s[ig * 16 + ii];
            double dd[DMAX], D2 = 0.0;
#pragma unroll
            for (int c = 0; c < DMAX; ++c) {
                const double df = xi[c] - xj[c];
                dd[c] = df * df;
                D2 += dd[c];
            }
            if (MODE == 0 && part.kind == GPX_PERIODIC) {
                const PeriodicGrad g =
                    periodic_grad(part.sf2, part.ell, part.period, D2);
                a_sf += t * g.g0;
                if (DMAX >= 2) {
                    a_e[0] += t * g.g1;
                    a_e[DMAX >= 2 ? 1 : 0] += t * g.g2;
                }
                continue;
            }
            const RadialGrad g = radial_grad_t<MODE == 0>(part.kind, part.two_logsf, part.sf2,
                                                          part.alpha, D2);
            a_sf += t * (2 * g.K);
            if (MODE == 0) a_x += t * g.xval;
            if (part.iso) {
                a_e[0] += t * g.isoval;
            } else {
                const double cf = g.zero ? 0.0 : t * (g.Mv / g.rdiv);
#pragma unroll
                for (int c = 0; c < DMAX; ++c) a_e[c] += cf * dd[c];
            }
        }
        // block reduction of this part's accumulators
        const int nh = part.nhyper;          // 1 + (#ell | 2 for periodic) (+1 RQ alpha)
        const int ne = part.kind == GPX_RQ ? nh - 2 : nh - 1;
        double v = wave_sum(a_sf);
        if (lane == 0) red[ig][0] = v;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < ne) {
                v = wave_sum(a_e[c]);
                if (lane == 0) red[ig][1 + c] = v;
            }
        if (part.kind == GPX_RQ) {
            v = wave_sum(a_x);
            if (lane == 0) red[ig][nh - 1] = v;
        }
        __syncthreads();
        if (tid < nh) {
            const double v4 = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            // the same thread wrote the slot of the earlier copy of a repeated primitive
            pout[1 + part.hoff + tid] = part.dup ? pout[1 + part.hoff + tid] + v4 : v4;
        }
    }
    __syncthreads();
    {
        double v = wave_sum(trq);
        if (lane == 0) red[ig][0] = v;
        __syncthreads();
        if (tid == 0) pout[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    }
}

// ---- trace_grad, row-persistent form (sums of SE / Matern parts) ---------------
// One workgroup owns a 64-row block of K^-1 and walks the upper tiles of that row
// with stride gridDim.x: x_i (scaled) and alpha_i are staged once, the D+2
// accumulators stay in registers across tiles and are reduced once per part. The
// kernel family is a template argument of the 16-pair loop, so an SE part pays for
// one exp per pair and no sqrt / division (the generic kernel above paid for the
// Matern guard division on every pair: 120 VALU instructions per pair against 56).
template <int DMAX, int KIND, bool ISO>
__device__ __forceinline__ void trace_pairs_t(const KPart &part, const double (*xi)[DMAX + 1],
                                              const double (&xj)[DMAX], const double (&wq)[16],
                                              double &a_sf, double (&a_e)[DMAX])
{
    const double tl = part.two_logsf;
    // (the body of a pair is one dependent chain -- eight FMAs into D2, then the exponential;
    // without the isotropic branch inside, the pairs of an unrolled group share a basic block
    // and the scheduler interleaves their chains)
    constexpr int UNROLL = DMAX <= 8 ? GPX_TRACE_UNROLL : 2;   // wider inputs: register budget
#pragma unroll UNROLL
    for (int ii = 0; ii < 16; ++ii) {
        const double t = wq[ii];
        double dd[DMAX], D2 = 0.0;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) {
            const double df = xi[ii][c] - xj[c];
            dd[c] = df * df;
            D2 += dd[c];
        }
        double cf, isoval;
        if (KIND == GPX_SE) {                              // se.py:57-66
            const double K = gpx_exp(tl - D2 / 2);
            cf = t * K;
            a_sf += cf;
            isoval = cf * D2;
        } else {                                           // matern.py:76-90
            const double r = sqrt(D2);
            const double S = gpx_exp(tl - r);
            const double f = KIND == GPX_MATERN1 ? 1.0
                             : (KIND == GPX_MATERN3 ? 1 + r : 1 + r * (1 + r / 3.));
            const double df = KIND == GPX_MATERN1 ? 1.0
                              : (KIND == GPX_MATERN3 ? r : r * (1 + r) / 3.);
            a_sf += t * (S * f);
            const double Mv = S * df;
            isoval = t * (Mv * r);
            cf = r < 1e-12 ? 0.0 : t * (Mv / r);
        }
        if (ISO) {
            a_e[0] += isoval;
        } else {
#pragma unroll
            for (int c = 0; c < DMAX; ++c) a_e[c] += cf * dd[c];
        }
    }
}

template <int DMAX, int KIND>
__device__ __forceinline__ void trace_pairs(const KPart &part, const double (*xi)[DMAX + 1],
                                            const double (&xj)[DMAX], const double (&wq)[16],
                                            double &a_sf, double (&a_e)[DMAX])
{
    if (part.iso != 0) trace_pairs_t<DMAX, KIND, true>(part, xi, xj, wq, a_sf, a_e);
    else trace_pairs_t<DMAX, KIND, false>(part, xi, xj, wq, a_sf, a_e);
}

// x / ell of every SE / Matern part, once per evaluation: Xs[p][r][c], r < np (rows >= n
// repeat row n - 1, as the kernels' clamps did), c < dmax (0 beyond d). The row kernel used
// to divide on the fly, per tile and per wave: the `c < d` guards made that a chain of
// load -> wait -> 12-instruction division blocks, eight dependent L2 round trips in front
// of every 16-pair body. Same division, same values.
template <bool MB>
__global__ __launch_bounds__(256) void xscale_kernel(KParams kp_arg, const double *__restrict__ X,
                                                     int n, int d, int np, int dmax,
                                                     double *__restrict__ Xs,
                                                     const MemberParams *__restrict__ mp,
                                                     long long pstride)
{
    const KParams &kp = MB ? mp[blockIdx.z].kp : kp_arg;
    if (MB) Xs += (long long)blockIdx.z * pstride;
    const int p = blockIdx.y;
    const KPart &part = kp.part[p];
    if (part.kind != GPX_SE && part.kind != GPX_MATERN1 && part.kind != GPX_MATERN3 &&
        part.kind != GPX_MATERN5)
        return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= np * dmax) return;
    const int r = e / dmax, c = e - r * dmax;
    const int gi = min(r, n - 1);
    Xs[(size_t)p * np * dmax + e] = c < d ? X[(size_t)gi * d + c] / part.scale[c] : 0.0;
}

// The launch handles the parts whose family is KIND (one launch per family present
// in the kernel, usually one); do_trq: this launch also owns slot 0, tr(Q).
// (Round 3: asking for more waves per SIMD through the launch bounds -- the D = 8
// instance then takes 102 instead of 192 VGPRs without spilling -- made it slower, 0.50
// -> 0.58 ms, and the D = 16 Matern instance spills: 1.6 -> 3.6 ms. The kernel lives on
// the instruction-level parallelism of its 16-row unrolled body, not on occupancy.)
// (the D = 16 instances need 257-259 VGPRs unconstrained: one register allocation step over
// two waves per SIMD; asked for two they fit)
template <int DMAX, int KIND, bool MB>
__global__ __launch_bounds__(256, (DMAX == 16 ? 2 : 1)) void trace_grad_rows_kernel(
    KParams kp_arg, const double *__restrict__ Xs, int n,
    const double *__restrict__ Kinv, int ld, const double *__restrict__ alpha,
    double *__restrict__ partial, int nacc, int do_trq, const MemberParams *__restrict__ mp,
    long long mstride, long long vstride, long long pstride)
{
    // (member-batched: blockIdx.z = member; its scaled inputs lie inside its scratch)
    const KParams &kp = MB ? mp[blockIdx.z].kp : kp_arg;
    if (MB) {
        Xs += (long long)blockIdx.z * pstride;
        Kinv += (long long)blockIdx.z * mstride;
        alpha += (long long)blockIdx.z * vstride;
        partial += (long long)blockIdx.z * pstride;
    }
    const int T = gridDim.y, C = gridDim.x;
    const int bi = blockIdx.y, c0 = blockIdx.x;
    double *pout = partial + ((size_t)bi * C + c0) * nacc;
    const int tid = threadIdx.x;
    if (bi + c0 >= T) {                          // no tile of this row for this chunk
        if (tid == 0 && do_trq) pout[0] = 0.0;
        for (int p = 0; p < kp.nparts; ++p)
            if (kp.part[p].kind == KIND && tid < kp.part[p].nhyper)
                pout[1 + kp.part[p].hoff + tid] = 0.0;
        return;
    }
    __shared__ double xi_s[KT][DMAX + 1];
    __shared__ double ai_s[KT];
    __shared__ double red[4][DMAX + 3];
    const int lane = tid & 63, ig = tid >> 6;
    const int i0 = bi * KT;
    if (tid < KT) ai_s[tid] = alpha[min(i0 + tid, n - 1)];
    double trq = 0.0;
    // rows i0 .. i0 + 63 of K^-1 (byte offsets inside: < 64 ld 8 + 8 np, far below 2 GB)
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Kinv + (size_t)i0 * ld), 0, 0x7fffffff, 0x00020000);

    bool first = true;
    for (int p = 0; p < kp.nparts; ++p) {
        const KPart &part = kp.part[p];
        if (part.kind != KIND) continue;
        // this part's inputs, scaled and padded by xscale_kernel: [np][DMAX]
        const double *__restrict__ xs = Xs + (size_t)p * gridDim.y * KT * DMAX;
        __syncthreads();
        for (int e = tid; e < KT * DMAX; e += 256) {
            const int r = e / DMAX, c = e - r * DMAX;
            xi_s[r][c] = xs[(size_t)(i0 + r) * DMAX + c];
        }
        __syncthreads();
        double a_sf = 0.0, a_e[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) a_e[c] = 0.0;

        // One tile's operands: 16 rows of K^-1 (this wave's rows, one column per lane),
        // alpha_j and the scaled x_j. The rows come through a buffer descriptor over this
        // workgroup's 64 rows with the row offset as the scalar offset: one address register
        // instead of sixteen 64-bit ones, which is what lets the NEXT tile's operands be in
        // flight during the 16-pair body (D <= 8: 191 -> 213 VGPRs, still two waves per
        // SIMD; with global_load addresses the same prefetch took 276 and was slower).
        struct TileOps {
            double q[16], aj, xj[DMAX];
        };
        const int rowoff = __builtin_amdgcn_readfirstlane(ig * 16 * ld * 8);
        auto fetch = [&](TileOps &o, int bj) {
            const int gj = bj * KT + lane;
#pragma unroll
            for (int ii = 0; ii < 16; ++ii)
                o.q[ii] = __builtin_bit_cast(
                    double, __builtin_amdgcn_raw_buffer_load_b64(rK, gj * 8, rowoff + ii * ld * 8, 0));
            o.aj = alpha[min(gj, n - 1)];
            const double2 *__restrict__ xp =
                reinterpret_cast<const double2 *>(xs + (size_t)gj * DMAX);
#pragma unroll
            for (int c = 0; c < DMAX; c += 2) {
                const double2 v = xp[c / 2];
                o.xj[c] = v.x;
                o.xj[c + 1] = v.y;
            }
        };
        constexpr bool AHEAD = GPX_TRACE_AHEAD && DMAX <= 8;
        TileOps cur, nxt;
        if (AHEAD) fetch(cur, bi + c0);
        for (int bj = bi + c0; bj < T; bj += C) {
            const int j0 = bj * KT;
            const int gj = j0 + lane;
            if (!AHEAD) fetch(cur, bj);
            else if (bj + C < T) fetch(nxt, bj + C);
            const double aj = cur.aj;
            const double (&q)[16] = cur.q;
            const double (&xj)[DMAX] = cur.xj;
            // weights: the full symmetric sum from the upper triangle (exact.py:129-138)
            double wq[16];
            if (bj > bi && i0 + KT <= n && j0 + KT <= n) {
#pragma unroll
                for (int ii = 0; ii < 16; ++ii)
                    wq[ii] = 2.0 * (q[ii] - ai_s[ig * 16 + ii] * aj);
            } else {
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) {
                    const int gi = i0 + ig * 16 + ii;
                    double w = 0.0;
                    if (gi < n && gj < n) w = gi < gj ? 2.0 : (gi == gj ? 1.0 : 0.0);
                    const double qq = w != 0.0 ? q[ii] - ai_s[ig * 16 + ii] * aj : 0.0;
                    if (first && gi == gj && w != 0.0) trq += qq;
                    wq[ii] = w * qq;
                }
            }
            trace_pairs<DMAX, KIND>(part, &xi_s[ig * 16], xj, wq, a_sf, a_e);
            if (AHEAD) cur = nxt;
        }
        first = false;
        // block reduction of this part's accumulators: [2 sum w q K | ell slots]
        const int nh = part.nhyper;
        double v = wave_sum(2.0 * a_sf);
        if (lane == 0) red[ig][0] = v;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < nh - 1) {
                v = wave_sum(a_e[c]);
                if (lane == 0) red[ig][1 + c] = v;
            }
        __syncthreads();
        if (tid < nh)
            pout[1 + part.hoff + tid] =
                red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
    __syncthreads();
    if (do_trq) {
        double v = wave_sum(trq);
        if (lane == 0) red[ig][0] = v;
        __syncthreads();
        if (tid == 0) pout[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    }
}

// deterministic second stage: acc[h] = sum over blocks of partial[b][h]
__global__ __launch_bounds__(256) void trace_reduce_kernel(
    const double *__restrict__ partial, int nblocks, int nacc, double *__restrict__ acc,
    long long pstride, int astride)
{
    partial += (long long)blockIdx.z * pstride;          // blockIdx.z = member
    acc += (long long)blockIdx.z * astride;
    __shared__ double red[256];
    const int h = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(size_t)b * nacc + h];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[h] = red[0];
}

size_t gpx_trace_scratch(int np)
{
    const size_t T = np / KT;
    // per-tile partial sums, then the scaled inputs of the row kernel (xscale_kernel)
    return T * T * (size_t)TG_MAXACC + (size_t)GPX_MAX_PARTS * np * GPX_MAX_DIM;
}

int gpx_trace_grad(hipStream_t s, const KParams &kp, const double *X, int n, int np,
                   int d, const double *Kinv, int ld, const double *alpha,
                   double *partial, double *acc, const MemberBatch *mb, int astride)
{
    // mb: member-batched (kp then only describes the STRUCTURE the members share: parts,
    // kinds, iso flags, slots; every member's scratch is gpx_trace_scratch(np) doubles, its
    // accumulators astride doubles behind those of the member before)
    const int T = np / KT;
    const int nacc = 1 + kp.nhyper;
    const int members = mb ? mb->count : 1;
    const MemberParams *mp = mb ? mb->params : nullptr;
    const long long mstride = mb ? mb->mstride : 0, vstride = mb ? mb->vstride : 0;
    const long long pstride = mb ? (long long)gpx_trace_scratch(np) : 0;
    dim3 grid(T, T, members);
    bool simple = kp.nprod == 0;
    for (int p = 0; p < kp.nparts; ++p)
        simple = simple && (kp.part[p].kind == GPX_SE || kp.part[p].kind == GPX_MATERN1 ||
                            kp.part[p].kind == GPX_MATERN3 || kp.part[p].kind == GPX_MATERN5);
    static const int rows_env = getenv("GPX_TRACE_ROWS") ? atoi(getenv("GPX_TRACE_ROWS")) : 16;
    if (simple && rows_env > 0) {
        // row-persistent kernel: C column chunks per 64-row block
        const int C = std::min(T, rows_env);
        dim3 rgrid(C, T, members);
        const int dmax = d <= 8 ? 8 : (d <= 16 ? 16 : 32);
        double *Xs = partial + (size_t)T * T * TG_MAXACC;
        const dim3 xgrid((np * dmax + 255) / 256, kp.nparts, members);
        if (mp)
            hipLaunchKernelGGL(xscale_kernel<true>, xgrid, dim3(256), 0, s, kp, X, n, d, np, dmax,
                               Xs, mp, pstride);
        else
            hipLaunchKernelGGL(xscale_kernel<false>, xgrid, dim3(256), 0, s, kp, X, n, d, np, dmax,
                               Xs, mp, pstride);
        GPX_HIP(hipGetLastError());
        bool trq_done = false;
        const int kinds[4] = {GPX_SE, GPX_MATERN1, GPX_MATERN3, GPX_MATERN5};
        for (int kind : kinds) {
            bool present = false;
            for (int p = 0; p < kp.nparts; ++p) present = present || kp.part[p].kind == kind;
            if (!present) continue;
            const int do_trq = trq_done ? 0 : 1;
            trq_done = true;
#define GPX_TR(DM, KD)                                                                       \
    do {                                                                                     \
        if (mp)                                                                              \
            hipLaunchKernelGGL((trace_grad_rows_kernel<DM, KD, true>), rgrid, dim3(256), 0, s, kp, \
                               Xs, n, Kinv, ld, alpha, partial, nacc, do_trq, mp, mstride,   \
                               vstride, pstride);                                            \
        else                                                                                 \
            hipLaunchKernelGGL((trace_grad_rows_kernel<DM, KD, false>), rgrid, dim3(256), 0, s, kp, \
                               Xs, n, Kinv, ld, alpha, partial, nacc, do_trq, mp, mstride,   \
                               vstride, pstride);                                            \
    } while (0)
#define GPX_TRD(KD)                                                                          \
    do {                                                                                     \
        if (d <= 8) GPX_TR(8, KD);                                                           \
        else if (d <= 16) GPX_TR(16, KD);                                                    \
        else GPX_TR(32, KD);                                                                 \
    } while (0)
            if (kind == GPX_SE) GPX_TRD(GPX_SE);
            else if (kind == GPX_MATERN1) GPX_TRD(GPX_MATERN1);
            else if (kind == GPX_MATERN3) GPX_TRD(GPX_MATERN3);
            else GPX_TRD(GPX_MATERN5);
#undef GPX_TRD
#undef GPX_TR
            GPX_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(trace_reduce_kernel, dim3(nacc, 1, members), dim3(256), 0, s, partial,
                           T * C, nacc, acc, pstride, astride);
        GPX_HIP(hipGetLastError());
        return 0;
    }
#define GPX_TG(DM, MODE)                                                                    \
    do {                                                                                    \
        if (mp)                                                                             \
            hipLaunchKernelGGL((trace_grad_kernel<DM, MODE, true>), grid, dim3(256), 0, s, kp, X, \
                               n, d, Kinv, ld, alpha, partial, nacc, mp, mstride, vstride,  \
                               pstride);                                                    \
        else                                                                                \
            hipLaunchKernelGGL((trace_grad_kernel<DM, MODE, false>), grid, dim3(256), 0, s, kp, X, \
                               n, d, Kinv, ld, alpha, partial, nacc, mp, mstride, vstride,  \
                               pstride);                                                    \
    } while (0)
    // periodic parts need two slots besides sf, so DMAX >= 2
    if (d <= 8) {
        if (simple) GPX_TG(8, 1); else GPX_TG(8, 0);
    } else if (d <= 16) {
        if (simple) GPX_TG(16, 1); else GPX_TG(16, 0);
    } else {
        if (simple) GPX_TG(32, 1); else GPX_TG(32, 0);
    }
#undef GPX_TG
    GPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(trace_reduce_kernel, dim3(nacc, 1, members), dim3(256), 0, s, partial,
                       T * T, nacc, acc, pstride, astride);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- input gradients -----------------------------------------------------------
// d k(x1, x2) / d x2 ("grady"; gradx is its negative) for one part, accumulated
// into g[0..d). SE: K (u1-u2)/ell (se.py:76-86); Matern: M (u1-u2)/ell with
// M = S df(r)/r guarded at r < 1e-12 (matern.py:100-114), u = x / ell;
// Periodic (one input dimension): 2 pi /(ell^2 p) K sin(2 (x1-x2) pi / p)
// (periodic.py:84-97).
template <int DMAX>
__device__ __forceinline__ void part_grady(const KPart &part, const double *__restrict__ x1,
                                           const double *__restrict__ x2, int d,
                                           double fac, double (&g)[DMAX])
{
    // fac: product of the other parts of a product group (_real.py:120-128)
    if (part.kind == GPX_PERIODIC) {
        const double D = (x1[0] - x2[0]) * M_PI / part.period;      // periodic.py:90-92
        const double sn = sin(D) / part.ell;
        const double K = part.sf2 * exp(-2 * (sn * sn));
        g[0] += fac * (2 * M_PI / (part.ell * part.ell) / part.period * K * sin(2 * D));
        return;
    }
    double u[DMAX];
    double D2 = 0.0;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
        if (c < d) {
            u[c] = x1[c] / part.scale[c] - x2[c] / part.scale[c];
            D2 += u[c] * u[c];
        }
    double cf;
    if (part.kind == GPX_RQ) {                         // rq.py:93-107
        const double E = 1 + 0.5 * D2 / part.alpha;
        cf = part.sf2 * pow(E, -part.alpha) / E;
    } else if (part.kind == GPX_SE) {
        cf = exp(part.two_logsf - D2 / 2);
    } else {
        const double r = sqrt(D2);
        const double S = exp(part.two_logsf - r);
        const double df = part.kind == GPX_MATERN1 ? 1.0
                          : (part.kind == GPX_MATERN3 ? r : r * (1 + r) / 3.);
        cf = r < 1e-12 ? 0.0 : S * df / r;
    }
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
        if (c < d) g[c] += fac * (cf * u[c] / part.scale[c]);
}

template <int DMAX>
__global__ __launch_bounds__(256) void kgrady_kernel(KParams kp, const double *__restrict__ X1,
                                                     int n1, const double *__restrict__ X2,
                                                     int n2, int d, double sign,
                                                     double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n2) return;
    double g[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) g[c] = 0.0;
    for (int p = 0; p < kp.nparts; ++p)
        part_grady<DMAX>(kp.part[p], X1 + (size_t)i * d, X2 + (size_t)j * d, d,
                         group_factor(kp, p, X1 + (size_t)i * d, X2 + (size_t)j * d, d), g);
    double *o = out + ((size_t)i * n2 + j) * d;
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
        if (c < d) o[c] = sign * g[c];
}

int gpx_kgrady(hipStream_t s, const KParams &kp, const double *X1, int n1, const double *X2,
               int n2, int d, double sign, double *out)
{
    for (int p = 0; p < kp.nparts; ++p)
        if (kp.part[p].kind == GPX_PERIODIC && d != 1) {
            gpx_set_error("input gradients of the periodic kernel need ndim == 1");
            return -1;
        }
    dim3 grid((n2 + 255) / 256, n1);
    if (d <= 8)
        hipLaunchKernelGGL(kgrady_kernel<8>, grid, dim3(256), 0, s, kp, X1, n1, X2, n2, d,
                           sign, out);
    else if (d <= 16)
        hipLaunchKernelGGL(kgrady_kernel<16>, grid, dim3(256), 0, s, kp, X1, n1, X2, n2, d,
                           sign, out);
    else
        hipLaunchKernelGGL(kgrady_kernel<32>, grid, dim3(256), 0, s, kp, X1, n1, X2, n2, d,
                           sign, out);
    GPX_HIP(hipGetLastError());
    return 0;
}

// dmu[m][c] = sum_i grady_c(x_i, xs_m) alpha_i            (exact.py:103-107, with
// ds2[m][c] = -2 sum_i grady_c(x_i, xs_m) beta[i][m]       R^-T dK . a = dK . alpha
// and R^-T dK . R^-T K = dK . K^-1 K: beta = K^-1 K(X, Xs), exact.py:109-110).
// Block = 16 test points x 16 row lanes; beta rows are read 128 B at a time.
template <int DMAX, bool MB>
__global__ __launch_bounds__(256) void posterior_grad_kernel(
    KParams kp_arg, const double *__restrict__ X, int n, const double *__restrict__ Xs, int m,
    int d, const double *__restrict__ alpha, const double *__restrict__ beta, int ldb,
    double *__restrict__ part, const MemberParams *__restrict__ mp, long long vstride,
    long long bstride, long long pstride)
{
    const KParams &kp = MB ? mp[blockIdx.z].kp : kp_arg;
    if (MB) {                                            // blockIdx.z = member
        alpha += (long long)blockIdx.z * vstride;
        beta += (long long)blockIdx.z * bstride;
        part += (long long)blockIdx.z * pstride;
    }
    // grid.y row chunks: with a few test points the rows are what fills the GPU;
    // every chunk writes its partial sums, posterior_grad_final_kernel adds them
    __shared__ double red[4][16][2 * DMAX];
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int wave = threadIdx.x >> 6;
    const int mj = blockIdx.x * 16 + c;
    const int mc = min(mj, m - 1);
    double xs[DMAX];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) xs[q] = q < d ? Xs[(size_t)mc * d + q] : 0.0;
    double am[DMAX], as2[DMAX];
#pragma unroll
    for (int q = 0; q < DMAX; ++q) am[q] = as2[q] = 0.0;
    const int rows = (n + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ibeg = (int)blockIdx.y * rows, iend = min(n, ibeg + rows);
    for (int i = ibeg + rl; i < iend; i += 16) {
        double g[DMAX];
#pragma unroll
        for (int q = 0; q < DMAX; ++q) g[q] = 0.0;
        for (int p = 0; p < kp.nparts; ++p)
            part_grady<DMAX>(kp.part[p], X + (size_t)i * d, xs, d,
                             group_factor(kp, p, X + (size_t)i * d, xs, d), g);
        const double ai = alpha[i];
        const double bi = beta[(size_t)i * ldb + mc];
#pragma unroll
        for (int q = 0; q < DMAX; ++q)
            if (q < d) {
                am[q] += g[q] * ai;
                as2[q] += g[q] * bi;
            }
    }
    // reduce over the 16 row lanes: 4 per wave (lanes c, c+16, c+32, c+48), 4 waves
#pragma unroll
    for (int q = 0; q < DMAX; ++q) {
        double v = am[q], w = as2[q];
        v += __shfl_down(v, 32, 64); w += __shfl_down(w, 32, 64);
        v += __shfl_down(v, 16, 64); w += __shfl_down(w, 16, 64);
        if ((threadIdx.x & 63) < 16) {
            red[wave][c][2 * q] = v;
            red[wave][c][2 * q + 1] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16 && mj < m) {
        double *po = part + ((size_t)blockIdx.y * m + mj) * 2 * d;
        for (int q = 0; q < d; ++q) {
            double v = 0.0, w = 0.0;
            for (int k = 0; k < 4; ++k) {
                v += red[k][c][2 * q];
                w += red[k][c][2 * q + 1];
            }
            po[2 * q] = v;
            po[2 * q + 1] = w;
        }
    }
}

__global__ __launch_bounds__(256) void posterior_grad_final_kernel(
    const double *__restrict__ part, int chunks, int m, int d, double *__restrict__ dmu,
    double *__restrict__ ds2, long long pstride)
{
    part += (long long)blockIdx.z * pstride;               // blockIdx.z = member
    dmu += (long long)blockIdx.z * m * d;
    ds2 += (long long)blockIdx.z * m * d;
    const int e = blockIdx.x * 256 + threadIdx.x;          // (test point, dimension)
    if (e >= m * d) return;
    double v = 0.0, w = 0.0;
    for (int k = 0; k < chunks; ++k) {
        v += part[((size_t)k * m * d + e) * 2];
        w += part[((size_t)k * m * d + e) * 2 + 1];
    }
    dmu[e] = v;
    ds2[e] = -2.0 * w;
}

// row chunks of gpx_posterior_grad for m test points on n training points
static int posterior_grad_chunks(int n, int m)
{
    const int col_blocks = (m + 15) / 16;
    int chunks = (1024 + col_blocks - 1) / col_blocks;      // ~1024 workgroups
    chunks = std::min(chunks, (n + 255) / 256);             // >= 16 rows per lane
    return std::max(1, std::min(chunks, 128));
}

size_t gpx_posterior_grad_scratch(int n, int m, int d)
{
    return (size_t)posterior_grad_chunks(n, m) * m * 2 * d;
}

int gpx_posterior_grad(hipStream_t s, const KParams &kp, const double *X, int n,
                       const double *Xs, int m, int d, const double *alpha,
                       const double *beta, int ldb, double *part, double *dmu, double *ds2,
                       const MemberBatch *mb, long long bstride)
{
    for (int p = 0; p < kp.nparts; ++p)
        if (kp.part[p].kind == GPX_PERIODIC && d != 1) {
            gpx_set_error("input gradients of the periodic kernel need ndim == 1");
            return -1;
        }
    const int chunks = posterior_grad_chunks(n, m);
    const int members = mb ? mb->count : 1;
    const MemberParams *mp = mb ? mb->params : nullptr;
    const long long vstride = mb ? mb->vstride : 0;
    const long long pstride = mb ? (long long)gpx_posterior_grad_scratch(n, m, d) : 0;
    dim3 grid((m + 15) / 16, chunks, members);
#define GPX_PG(DM)                                                                           \
    do {                                                                                     \
        if (mp)                                                                              \
            hipLaunchKernelGGL((posterior_grad_kernel<DM, true>), grid, dim3(256), 0, s, kp, X, n, \
                               Xs, m, d, alpha, beta, ldb, part, mp, vstride, bstride, pstride); \
        else                                                                                 \
            hipLaunchKernelGGL((posterior_grad_kernel<DM, false>), grid, dim3(256), 0, s, kp, X, n, \
                               Xs, m, d, alpha, beta, ldb, part, mp, vstride, bstride, pstride); \
    } while (0)
    if (d <= 8) GPX_PG(8);
    else if (d <= 16) GPX_PG(16);
    else GPX_PG(32);
#undef GPX_PG
    hipLaunchKernelGGL(posterior_grad_final_kernel, dim3((m * d + 255) / 256, 1, members),
                       dim3(256), 0, s, part, chunks, m, d, dmu, ds2, pstride);
    GPX_HIP(hipGetLastError());
    return 0;
}
