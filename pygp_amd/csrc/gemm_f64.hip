// fp64 tile engine for gfx950: C = alpha * op(A) * op(B) + beta * C on
// v_mfma_f64_16x16x4_f64.
//
// One 256-thread workgroup (4 waves, 2x2) owns a TILE x TILE block of C,
// TILE = 128 (each wave 64x64 = 4x4 MFMA tiles, 128 accumulator VGPRs) or, for
// launches that would leave most of the 256 CUs idle, TILE = 64 (each wave
// 32x32 = 2x2 MFMA tiles; 4x the workgroups, 1/4 of the serial K-loop latency
// per workgroup). The K loop walks BK=16 slices: global -> registers (16-B
// loads, one slice ahead, issued before the MFMAs of the current slice) -> LDS
// (double buffered) -> one ds_read_b64 per fragment -> MFMA.
//
// LDS images keep the operand's own storage order so that the staging writes
// are 16-B and conflict-free either way:
//   k-major operand  (stored [k][mn]):  S[k][mn], row stride TILE+16 doubles
//   mn-major operand (stored [mn][k]):  S[mn][k], row stride 18 doubles
// both strides put the two k-rows / sixteen mn-rows a 32-lane group reads on
// disjoint banks (ds_read_b64 banks = (addr/4) % 64).
//
// The f64 MFMA is slow in wall-clock terms (~30 ns per instruction per SIMD,
// tools/probe_mfma.hip), so a slice costs a wave 64 MFMAs = 1.9 us at TILE=128;
// LDS and global traffic are far off the critical path. What matters is that
// all four SIMDs of every CU always have an MFMA to issue.
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

#include "gemm_tile.h"

template <int TA, int TB, typename G>
__global__ __launch_bounds__(G::NTH, G::MINW) void gemm_f64_kernel(GemmArgs g)
{
    constexpr int TILE = G::TILE, WTM = G::WTM, WTN = G::WTN;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *smem = reinterpret_cast<double *>(smem_raw);

    // ---- block id -> output tile -------------------------------------------
    // Workgroups are dealt round-robin to the 8 XCDs (block b and b+8 share an
    // XCD and its private 4 MB L2). When the grid allows it, every XCD walks
    // whole 8x8 macro tiles: the 64 workgroups it runs at a time then share 8
    // row panels and 8 column panels in its L2 instead of streaming 64+ panels
    // from HBM (TCC hit rate of the lauum launch 22% -> see DESIGN.md). Macro
    // tiles themselves are dealt round-robin to XCDs, in g.order, so triangular
    // workloads stay balanced across XCDs. Placement only affects speed.
    int tm = blockIdx.y, tn = blockIdx.x;
    if (g.tiles) {
        // structured launches walk a host-built list of live tiles, longest
        // k-range first: no dead workgroups (each would still have to wait for
        // a free 72 KB LDS slot before it could exit) and an LPT schedule
        tm = g.tiles[2 * blockIdx.x];
        tn = g.tiles[2 * blockIdx.x + 1];
    } else {
        const int gx = gridDim.x, gy = gridDim.y;
        const int mgx = gx >> 3, mgy = gy >> 3;
        const bool macro = g.swizzle && !(gx & 7) && !(gy & 7) && !((mgx * mgy) & 7);
        const int lin = blockIdx.y * gx + blockIdx.x;
        int u = lin, ux = gx, uy = gy;           // walk position and its extents
        int w = 0;
        if (macro) {
            const int xcd = lin & 7, q = lin >> 3;
            u = (q >> 6) * 8 + xcd;              // macro tile id
            w = q & 63;                          // tile inside the macro tile
            ux = mgx;
            uy = mgy;
        }
        int um, un;
        if (g.order == 0) {
            um = u / ux; un = u - um * ux;
        } else if (g.order == 1) {
            um = u / ux; un = u - um * ux; um = uy - 1 - um;
        } else {
            un = u / uy; um = u - un * uy;
            if (g.order == 2) un = ux - 1 - un;
        }
        if (macro) {
            tm = um * 8 + (w >> 3);
            tn = un * 8 + (w & 7);
        } else {
            tm = um;
            tn = un;
        }
    }
    const int m0 = tm * TILE, n0 = tn * TILE;
    if ((g.flags & GEMM_UPPER_ONLY) && n0 + TILE <= m0) return;
    int klo = 0, khi = g.K;
    if (g.flags & GEMM_KLO_M) klo = max(klo, m0 - g.kshift);
    if (g.flags & GEMM_KHI_M) khi = min(khi, m0 + TILE);
    if (g.flags & GEMM_KLO_N) klo = max(klo, n0 - g.kshift);
    if (g.flags & GEMM_KHI_N) khi = min(khi, n0 + TILE);
    int zb = blockIdx.z, zm = 0;         // batch index of the strides / member of a split
    if (g.kchunk > 0) {                 // split-K: this batch index owns one k chunk
        if (g.nsplit > 0) {
            zm = zb / g.nsplit;
            zb -= zm * g.nsplit;
        }
        klo = max(klo, zb * g.kchunk);
        khi = min(khi, (zb + 1) * g.kchunk);
    }
    klo &= ~(BK - 1);

    const double *__restrict__ A = g.A + (long long)zb * g.strideA + (long long)zm * g.mstrideA;
    const double *__restrict__ B = g.B + (long long)zb * g.strideB + (long long)zm * g.mstrideB;
    double *__restrict__ C = g.C + (long long)zb * g.strideC + (long long)zm * g.mstrideC;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int lr = lane & 15, lk = lane >> 4;

    // op(A)[m][k]: TA == 0 -> A[m*lda + k] (mn-major), TA == 1 -> A[k*lda + m]
    // op(B)[k][n]: TB == 0 -> B[k*ldb + n] (k-major),  TB == 1 -> B[n*ldb + k]
    constexpr bool AKM = (TA == 1);
    constexpr bool BKM = (TB == 0);
    // LDS strides of one k-step (4 k) and one MFMA tile (16 mn) per operand
    constexpr int AK = AKM ? 4 * G::KSTR : 4, AT = AKM ? 16 : 16 * MNSTR;
    constexpr int BKS = BKM ? 4 * G::KSTR : 4, BT = BKM ? 16 : 16 * MNSTR;

    // beta C enters as the initial accumulator (scaled by 1 / alpha): the tile of C is
    // requested together with the first operand slices, and the epilogue is stores only.
    // With the read at the end every workgroup of a launch of equal-k tiles (the rank-NB
    // updates of the factorisation) sat in a load-wait epilogue at the same time.
    const double alpha = g.alpha;
    const double beta = (g.beta0_from >= 0 && n0 >= g.beta0_from) ? 0.0 : g.beta;
    if (g.C2 && (m0 >> 7) != (n0 >> 7))                // off-diagonal tile of a diagonal block
        C = g.C2 + (long long)zb * g.strideC2;
    v4d acc[WTM][WTN];
    if (beta != 0.0) {
        const double sc = beta / alpha;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int col = n0 + wn * (TILE / G::WN) + j * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * (TILE / G::WM) + i * 16 + lk + 4 * r;
                    acc[i][j][r] = sc * C[(size_t)row * g.ldc + col];
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    const int nslice = (khi - klo) / BK;
    if (nslice > 0) {
        double *As = smem, *Bs = smem + 2 * G::OPER;       // [2][OPER] each
        // GEMM_KREV walks k downwards: tiles whose k-ranges share their upper
        // end (k >= tile start) then read the same slices at the same time
        const bool rev = (g.flags & GEMM_KREV) != 0;
        const int kfirst = rev ? khi - BK : klo, kstep = rev ? -BK : BK;
        const int amn = wm * (TILE / G::WM) + lr, bmn = wn * (TILE / G::WN) + lr;
        const double *ap0 = As + (AKM ? lk * G::KSTR + amn : amn * MNSTR + lk);
        const double *bp0 = Bs + (BKM ? lk * G::KSTR + bmn : bmn * MNSTR + lk);
        if constexpr (!G::DEEP) {
            Regs<G::NLOAD> ra, rb;
            ra = load_slice<G, AKM>(A, g.lda, m0, kfirst, tid);
            rb = load_slice<G, BKM>(B, g.ldb, n0, kfirst, tid);
            store_slice<G, AKM>(As, tid, ra);
            store_slice<G, BKM>(Bs, tid, rb);
            __syncthreads();
            // steady state: prefetch slice s+1 into registers (unconditionally,
            // so that ra/rb stay in VGPRs -- a conditional prefetch sends them
            // to scratch), run the MFMAs of slice s, then publish s+1 to LDS
            for (int s = 0; s + 1 < nslice; ++s) {
                const int cur = s & 1;
                const int k0 = kfirst + (s + 1) * kstep;
                ra = load_slice<G, AKM>(A, g.lda, m0, k0, tid);
                rb = load_slice<G, BKM>(B, g.ldb, n0, k0, tid);
                mfma_slice<WTM, WTN, AK, AT, BKS, BT>(ap0 + cur * G::OPER,
                                                      bp0 + cur * G::OPER, acc);
                const int nxt = cur ^ 1;
                store_slice<G, AKM>(As + nxt * G::OPER, tid, ra);
                store_slice<G, BKM>(Bs + nxt * G::OPER, tid, rb);
                __syncthreads();
            }
            const int cur = (nslice - 1) & 1;
            mfma_slice<WTM, WTN, AK, AT, BKS, BT>(ap0 + cur * G::OPER, bp0 + cur * G::OPER,
                                                  acc);
        } else {
            // two slices in flight: registers r0 / r1 alternate, slice s+2 is
            // requested before the MFMAs of slice s and stored to LDS one whole
            // iteration later, so a load has two slice-times to come back
            Regs<G::NLOAD> ra0, rb0, ra1, rb1;
            const int last = nslice - 1;
            ra0 = load_slice<G, AKM>(A, g.lda, m0, kfirst, tid);
            rb0 = load_slice<G, BKM>(B, g.ldb, n0, kfirst, tid);
            {
                const int k1 = kfirst + min(1, last) * kstep;
                ra1 = load_slice<G, AKM>(A, g.lda, m0, k1, tid);
                rb1 = load_slice<G, BKM>(B, g.ldb, n0, k1, tid);
            }
            store_slice<G, AKM>(As, tid, ra0);
            store_slice<G, BKM>(Bs, tid, rb0);
            __syncthreads();
            // k-ranges are whole tiles, so nslice is even: no tail (a conditional
            // tail would demote the prefetch registers to scratch)
            for (int s = 0; s < nslice; s += 2) {
                {   // LDS[0] = slice s, r1 = slice s+1
                    const int k2 = kfirst + min(s + 2, last) * kstep;
                    ra0 = load_slice<G, AKM>(A, g.lda, m0, k2, tid);
                    rb0 = load_slice<G, BKM>(B, g.ldb, n0, k2, tid);
                    mfma_slice<WTM, WTN, AK, AT, BKS, BT>(ap0, bp0, acc);
                    store_slice<G, AKM>(As + G::OPER, tid, ra1);
                    store_slice<G, BKM>(Bs + G::OPER, tid, rb1);
                    __syncthreads();
                }
                {   // LDS[1] = slice s+1, r0 = slice s+2
                    const int k3 = kfirst + min(s + 3, last) * kstep;
                    ra1 = load_slice<G, AKM>(A, g.lda, m0, k3, tid);
                    rb1 = load_slice<G, BKM>(B, g.ldb, n0, k3, tid);
                    mfma_slice<WTM, WTN, AK, AT, BKS, BT>(ap0 + G::OPER, bp0 + G::OPER, acc);
                    store_slice<G, AKM>(As, tid, ra0);
                    store_slice<G, BKM>(Bs, tid, rb0);
                    __syncthreads();
                }
            }
        }
    }

    // epilogue. f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
    // (verified by tools/probe_mfma.hip)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (TILE / G::WM) + i * 16 + lk + 4 * r;
                const int col = n0 + wn * (TILE / G::WN) + j * 16 + lr;
                C[(size_t)row * g.ldc + col] = alpha * acc[i][j][r];
            }
}

typedef Geo<128, 2, 2> Big4;      // 256 threads, wave 64x64, two workgroups per CU
typedef Geo<128, 2, 4> Big8;      // 512 threads, wave 64x32, one workgroup per CU
typedef Geo<64, 2, 2> Small4;     // 256 threads, wave 32x32
typedef Geo<64, 2, 4> Small8;     // 512 threads, wave 32x16
typedef Geo<128, 2, 4, true> Big8D;   // Big8 with two slices in flight
typedef Geo<64, 2, 4, true> Small8D;

// ---- live-tile lists for structured launches ---------------------------------
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <cstring>
#include <mutex>
#include <tuple>
#include <vector>

struct TileList {
    int *dev = nullptr;
    int count = 0;
};

// Evaluations running side by side on several streams of one device (gpx_loglik_batch,
// gpx_posterior_batch) raise that device's count; it selects the tile order (see
// tile_order). Per device: a batch on one GPU says nothing about the others (the
// in-library multi-GPU path drives every device from its own host thread).
#define GPX_MAX_DEVICES 64
static std::atomic<int> g_concurrent[GPX_MAX_DEVICES];
void gpx_gemm_concurrency(int device, int delta)
{
    if (device >= 0 && device < GPX_MAX_DEVICES) g_concurrent[device] += delta;
}
int gpx_gemm_concurrent(int device)
{
    return device >= 0 && device < GPX_MAX_DEVICES ? g_concurrent[device].load() : 0;
}

// Tile order of structured launches. One evaluation at a time: XCD-aware (8x8 macro
// tiles per L2; the K^-1 launch runs 2.4% faster inside an evaluation). Several streams
// at once: plain longest-first, which measured 1% better there. GPX_TILE_XCD=0/1 forces.
// Read ONCE per gpx_gemm call: the two dispatches of a whole-rounds + remainder split
// must cut the same list, whatever another thread does to the count in between.
static int tile_order(int device)
{
    static const int xcd_env = getenv("GPX_TILE_XCD") ? atoi(getenv("GPX_TILE_XCD")) : -1;
    return xcd_env >= 0 ? xcd_env : (gpx_gemm_concurrent(device) > 0 ? 0 : 1);
}

// part 0: every live tile. Equal-k launches that do not fill whole rounds of `slots`
// workgroups are cut in two: part 1 = the first floor(L / slots) * slots 128-tiles,
// part 2 = the remaining 128-tiles as 64-tiles (tile must be 64 then; a quarter of the
// time each, so the last partial round costs a quarter too).
static int tile_list(int device, int xcd_order, int tile, int Tm, int Tn, int K, int flags,
                     TileList *out, int part = 0, int slots = 0, int kshift = 0)
{
    typedef std::tuple<int, int, int, int, int, int, int, int> Key;
    static std::map<Key, TileList> cache;
    static std::mutex mu;
    const int sflags = flags & (GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KHI_M | GEMM_KLO_N |
                                GEMM_KHI_N);
    const Key key(device, tile, Tm, Tn, K,
                  sflags | (xcd_order ? 1 << 20 : 0) | ((kshift / 64) << 5), part, slots);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return 0;
    }
    struct Item { int w, m, n; };
    std::vector<Item> items;
    items.reserve((size_t)Tm * Tn);
    // parts 1 and 2 are cut from the list of 128-tiles (Tm, Tn count 128-tiles then)
    const int gen_tile = part ? 128 : tile;
    for (int m = 0; m < Tm; ++m)
        for (int n = 0; n < Tn; ++n) {
            const int m0 = m * gen_tile, n0 = n * gen_tile;
            if ((sflags & GEMM_UPPER_ONLY) && n0 + gen_tile <= m0) continue;
            int klo = 0, khi = K;
            if (sflags & GEMM_KLO_M) klo = std::max(klo, m0 - kshift);
            if (sflags & GEMM_KHI_M) khi = std::min(khi, m0 + gen_tile);
            if (sflags & GEMM_KLO_N) klo = std::max(klo, n0 - kshift);
            if (sflags & GEMM_KHI_N) khi = std::min(khi, n0 + gen_tile);
            items.push_back({std::max(0, khi - klo), m, n});
        }
    // few macro tiles cannot be dealt evenly to 8 XCDs (the K^-1 launch at N = 4096 has
    // 10 of them: 0.87 instead of 0.55 ms): longest-first below 32 macro tiles
    if (xcd_order > 0 && items.size() >= 32 * 64) {
        // XCD-aware order: workgroup i runs on XCD i % 8, each with its
        // own L2. Deal 8x8 macro tiles (64 tiles sharing 8 row and 8 column panels)
        // to the XCDs, heaviest first to the least loaded, and interleave the 8
        // per-XCD sequences so that list position i belongs to XCD i % 8.
        const int MB = 8;
        const int Mm = (Tm + MB - 1) / MB, Mn = (Tn + MB - 1) / MB;
        struct Macro { long long w; std::vector<Item> t; };
        std::vector<Macro> macros((size_t)Mm * Mn);
        for (const Item &it : items) {
            Macro &mc = macros[(size_t)(it.m / MB) * Mn + it.n / MB];
            mc.w += it.w;
            mc.t.push_back(it);
        }
        std::vector<int> idx;
        for (size_t i = 0; i < macros.size(); ++i)
            if (!macros[i].t.empty()) idx.push_back((int)i);
        std::stable_sort(idx.begin(), idx.end(),
                         [&](int a, int b) { return macros[a].w > macros[b].w; });
        std::vector<Item> seq[8];
        long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int id : idx) {
            int x = 0;
            for (int k = 1; k < 8; ++k)
                if (load[k] < load[x]) x = k;
            load[x] += macros[id].w;
            std::vector<Item> &t = macros[id].t;
            std::stable_sort(t.begin(), t.end(),
                             [](const Item &a, const Item &b) { return a.w > b.w; });
            seq[x].insert(seq[x].end(), t.begin(), t.end());
        }
        std::vector<Item> inter;
        inter.reserve(items.size());
        size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        while (inter.size() < items.size()) {
            for (int x = 0; x < 8; ++x) {
                if (pos[x] < seq[x].size()) {
                    inter.push_back(seq[x][pos[x]++]);
                } else {
                    // keep the position -> XCD mapping: borrow from the longest rest
                    int y = 0;
                    for (int k = 1; k < 8; ++k)
                        if (seq[k].size() - pos[k] > seq[y].size() - pos[y]) y = k;
                    if (pos[y] < seq[y].size()) inter.push_back(seq[y][pos[y]++]);
                }
            }
        }
        items.swap(inter);
    } else {
        std::stable_sort(items.begin(), items.end(),
                         [](const Item &a, const Item &b) { return a.w > b.w; });
    }
    if (part) {
        const size_t whole = slots > 0 ? items.size() / slots * slots : items.size();
        if (part == 1) {
            items.resize(whole);
        } else {
            std::vector<Item> rest;
            for (size_t i = whole; i < items.size(); ++i)
                for (int q = 0; q < 4; ++q) {
                    const int m = 2 * items[i].m + (q >> 1), n = 2 * items[i].n + (q & 1);
                    if ((sflags & GEMM_UPPER_ONLY) && n < m) continue;   // below the diagonal
                    rest.push_back({items[i].w, m, n});
                }
            items.swap(rest);
        }
    }
    std::vector<int> flat(items.size() * 2);
    for (size_t i = 0; i < items.size(); ++i) {
        flat[2 * i] = items[i].m;
        flat[2 * i + 1] = items[i].n;
    }
    TileList tl;
    tl.count = (int)items.size();
    if (tl.count > 0) {
        GPX_HIP(hipMalloc((void **)&tl.dev, flat.size() * sizeof(int)));
        GPX_HIP(hipMemcpy(tl.dev, flat.data(), flat.size() * sizeof(int),
                          hipMemcpyHostToDevice));
    GPX_HIP(hipDeviceSynchronize());      // in HBM before any stream reads it
    }
    cache[key] = tl;
    *out = tl;
    return 0;
}

// GPX_GEMM_LOG=<file>: one line per kernel launch, in program order (developer aid:
// tools/gemm_trace_join.py matches them with a rocprofv3 kernel trace)
static void log_launch(hipStream_t s, int ta, int tb, int tile, const GemmArgs &g, int part,
                       int wgs)
{
    static FILE *f = nullptr;
    static bool init = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!init) {
        init = true;
        const char *e = getenv("GPX_GEMM_LOG");
        if (e && *e) f = fopen(e, "w");
    }
    if (!f) return;
    fprintf(f, "%p %d %d %d %d %d %d %d %d %d %d %g\n", (void *)s, ta, tb, tile, g.M, g.N, g.K,
            g.flags, g.kshift, part, wgs, g.beta);
    fflush(f);
}

// what one gpx_gemm call fixes for all of its dispatches
struct LaunchCtx {
    int device;
    int xcd_order;
};

template <int TA, int TB, typename G>
static int launch(hipStream_t s, const GemmArgs &g0, const LaunchCtx &lc, int part = 0)
{
    GemmArgs g = g0;
    const int structure = g.flags & (GEMM_UPPER_ONLY | GEMM_KLO_M | GEMM_KHI_M |
                                     GEMM_KLO_N | GEMM_KHI_N);
    g.tiles = nullptr;
    dim3 grid(g.N / G::TILE, g.M / G::TILE, g.batch > 0 ? g.batch : 1);
    if (part) {
        TileList tl;
        GPX_TRY(tile_list(lc.device, lc.xcd_order, G::TILE, g.M / 128, g.N / 128, g.K, g.flags,
                          &tl, part, g.slots > 0 ? g.slots : 512));
        if (tl.count == 0) return 0;
        g.tiles = tl.dev;
        grid = dim3(tl.count, 1, 1);
    } else if (structure && g.use_lists) {
        TileList tl;
        GPX_TRY(tile_list(lc.device, lc.xcd_order, G::TILE, g.M / G::TILE, g.N / G::TILE, g.K,
                          g.flags, &tl, 0, 0, g.kshift));
        if (tl.count == 0) return 0;
        g.tiles = tl.dev;
        grid = dim3(tl.count, 1, g.batch > 0 ? g.batch : 1);
    }
    log_launch(s, TA, TB, G::TILE, g, part, (int)(grid.x * grid.y * grid.z));
    hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, G>), grid, dim3(G::NTH), G::LDS_BYTES, s,
                       g);
    GPX_HIP(hipGetLastError());
    return 0;
}

template <int TA, int TB, typename G>
static int set_attr()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f64_kernel<TA, TB, G>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    return 0;
}

template <typename G>
static int set_attr_all()
{
    GPX_TRY((set_attr<0, 0, G>()));
    GPX_TRY((set_attr<0, 1, G>()));
    GPX_TRY((set_attr<1, 0, G>()));
    GPX_TRY((set_attr<1, 1, G>()));
    return 0;
}

int gpx_gemm_init()
{
    GPX_TRY(set_attr_all<Big4>());
    GPX_TRY(set_attr_all<Big8>());
    GPX_TRY(set_attr_all<Small4>());
    GPX_TRY(set_attr_all<Small8>());
    GPX_TRY(set_attr_all<Big8D>());
    GPX_TRY(set_attr_all<Small8D>());
    return 0;
}

template <typename G>
static int dispatch(hipStream_t s, int ta, int tb, const GemmArgs &g, const LaunchCtx &lc,
                    int part = 0)
{
    if (ta == 0 && tb == 0) return launch<0, 0, G>(s, g, lc, part);
    if (ta == 0 && tb == 1) return launch<0, 1, G>(s, g, lc, part);
    if (ta == 1 && tb == 0) return launch<1, 0, G>(s, g, lc, part);
    return launch<1, 1, G>(s, g, lc, part);
}

static int env_choice(const char *name)
{
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

__global__ void gpx_jitter_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int gpx_test_jitter(hipStream_t s)
{
    static const char *env = getenv("GPX_TEST_JITTER");
    if (!env) return 0;
    static std::mutex mu;
    static unsigned long long state = 0;
    static int max_us = 300;
    std::lock_guard<std::mutex> lock(mu);
    if (state == 0) {
        state = 0x9E3779B97F4A7C15ull ^ (unsigned long long)atoll(env);
        const char *c = strchr(env, ':');
        if (c) max_us = atoi(c + 1);
    }
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    const unsigned r = (unsigned)(state >> 33);
    if (r & 1) return 0;
    const long long ticks = (long long)((r >> 1) % (unsigned)(max_us + 1)) * 100;   // 100 MHz
    hipLaunchKernelGGL(gpx_jitter_kernel, dim3(1), dim3(64), 0, s, ticks);
    GPX_HIP(hipGetLastError());
    return 0;
}

int gpx_gemm(hipStream_t s, int ta, int tb, const GemmArgs &g)
{
    if (g.M <= 0 || g.N <= 0) return 0;
    GPX_TRY(gpx_test_jitter(s));
    if (g.M % GPX_TILE || g.N % GPX_TILE || g.K % (2 * BK) || g.lda % 2 || g.ldb % 2) {
        gpx_set_error("gpx_gemm: unpadded operands M=%d N=%d K=%d lda=%d ldb=%d", g.M,
                      g.N, g.K, g.lda, g.ldb);
        return -1;
    }
    LaunchCtx lc;
    GPX_HIP(hipGetDevice(&lc.device));
    lc.xcd_order = tile_order(lc.device);
    // live 128-tiles of this launch; below ~one per CU the 64-tile variants
    // quarter the serial K-loop latency of each workgroup
    long long tiles = (long long)(g.M / 128) * (g.N / 128) * (g.batch > 0 ? g.batch : 1);
    if (g.flags & GEMM_UPPER_ONLY) tiles = tiles / 2 + g.M / 256;
    static const int big_cfg = env_choice("GPX_GEMM_BIG");       // 4 or 8 (waves)
    static const int small_cfg = env_choice("GPX_GEMM_SMALL");
    static const int small_below = env_choice("GPX_GEMM_SMALL_BELOW");
    const int threshold = small_below > 0 ? small_below : 400;
    int tile = g.tile;
    if (tile == 0) tile = tiles < threshold ? 64 : 128;
    // Structured launches whose longest tile alone outlasts an even share of the work (the
    // block column of the inverse for a 1024-block: 56 x 8 tiles with k-ranges up to 7168
    // on 512 slots -- the launch took as long as its longest tile, 1.24 ms for 0.75 ms of
    // work): 64-tiles quarter the longest one. Estimated from the k-ranges, in units of
    // one 128-tile k-step; 64-tiles are charged 8 % for their lower efficiency.
    static const int balance_on = env_choice("GPX_GEMM_NOBALANCE") ? 0 : 1;
    const int kstruct_ = g.flags & (GEMM_KLO_M | GEMM_KHI_M | GEMM_KLO_N | GEMM_KHI_N);
    if (balance_on && g.tile == 0 && tile == 128 && kstruct_ && g.use_lists && g.batch <= 1 &&
        g.kchunk == 0) {
        long long work = 0, longest = 0;
        for (int m0 = 0; m0 < g.M; m0 += 128)
            for (int n0 = 0; n0 < g.N; n0 += 128) {
                if ((g.flags & GEMM_UPPER_ONLY) && n0 + 128 <= m0) continue;
                int klo = 0, khi = g.K;
                if (g.flags & GEMM_KLO_M) klo = std::max(klo, m0 - g.kshift);
                if (g.flags & GEMM_KHI_M) khi = std::min(khi, m0 + 128);
                if (g.flags & GEMM_KLO_N) klo = std::max(klo, n0 - g.kshift);
                if (g.flags & GEMM_KHI_N) khi = std::min(khi, n0 + 128);
                const long long len = std::max(0, khi - klo);
                work += len;
                longest = std::max(longest, len);
            }
        const double S = g.slots > 0 ? g.slots : 512;
        const double t128 = std::max((double)longest, work / S);
        const double t64 = std::max(longest / 4.0, 1.08 * work / S);
        if (t64 < 0.9 * t128) tile = 64;
    }
    // equal-k launches (rank-k updates, rectangular products): whole rounds of 128-tiles,
    // the rest as 64-tiles, when that is cheaper than a last round that is mostly empty
    // (a row panel of 896 tiles on 512 slots: 1.75 instead of 2 tile times)
    static const int split_on = env_choice("GPX_GEMM_NOSPLIT") ? 0 : 1;
    const int kstruct = g.flags & (GEMM_KLO_M | GEMM_KHI_M | GEMM_KLO_N | GEMM_KHI_N);
    if (split_on && !g.overlap && g.tile == 0 && tile == 128 && !kstruct && g.use_lists &&
        g.batch <= 1 && g.kchunk == 0 && !g.waves && !big_cfg && !small_cfg) {
        const long long S = g.slots > 0 ? g.slots : 512;
        long long L = (long long)(g.M / 128) * (g.N / 128);
        if (g.flags & GEMM_UPPER_ONLY) {
            L = 0;
            for (int m = 0; m < g.M / 128; ++m) L += std::max(0, g.N / 128 - m);
        }
        const long long rem = L % S;
        if (L > S && rem != 0) {
            const double plain = (double)((L + S - 1) / S);
            const double split = (double)(L / S) + (double)((4 * rem + S - 1) / S) / 4.0 + 0.03;
            if (split < plain) {
                GPX_TRY(dispatch<Big8D>(s, ta, tb, g, lc, 1));
                return dispatch<Small8D>(s, ta, tb, g, lc, 2);
            }
        }
    }
    if (tile == 64) {
        const int sw = g.waves ? g.waves : small_cfg;
        if (sw == 4) return dispatch<Small4>(s, ta, tb, g, lc);
        if (sw == 8) return dispatch<Small8>(s, ta, tb, g, lc);
        return dispatch<Small8D>(s, ta, tb, g, lc);
    }
    // tile == 128: the in-place panel multiply of trsm relies on one workgroup
    // per 128-row block, which both big configurations provide. Measured on
    // MI355X (tools/gemm_exp.py, interleaved): the 8-wave shape wins for every
    // operand layout once dead tiles are gone (TN 61 vs 51, NN 70 vs 67, NT 55 vs
    // 43 TFLOP/s at n = 8192).
    const int bw = g.waves ? g.waves : big_cfg;
    const bool use4 = bw == 4;
    if (use4) return dispatch<Big4>(s, ta, tb, g, lc);
    if (bw == 8) return dispatch<Big8>(s, ta, tb, g, lc);
    return dispatch<Big8D>(s, ta, tb, g, lc);     // default: 8 waves, two slices in flight
}
