// Diagonal-panel kernel: R = chol(A) (upper) and W = R^-1 of one n x n diagonal
// block, n = 128 T <= 1024, in ONE launch.
//
// Below ~1024 the recursive driver in chol.hip is a chain of small dependent
// launches (leaf, R12, SYRK, two inverse products per tree node: ~36 per
// 1024-block, each a few microseconds of work behind a dispatch). Here the same
// arithmetic is cut into tile tasks -- the 128x128 leaf (leaf_dev.h) and 64x64
// MFMA products (gemm_tile.h) -- that a small resident set of workgroups claims
// from a queue and synchronises with device-scope counters instead of kernel
// boundaries:
//
//   F(s)      leaf of tile (s,s): R_ss, W_ss                    [after all S(.,s,s)]
//   P(s,t)    R_st = W_ss^T X_st    (X: the staging area the off-diagonal tiles of the
//                                    block live in until their row panel, chol.hip)
//   S(j,s,t)  X_st -= R_js^T R_jt   (diagonal tiles: A_ss -= ...)
//   I1(i,j)   T_ij = sum_{k=i..j-1} W_ik R_kj                   (T in the scratch X)
//   I2(i,j)   W_ij = -T_ij W_jj
//
// Round 2 (default, GPX_PANEL_STREAM=1) takes W_ss, two hand-offs and a tile round
// trip off the chain of the diagonal tiles:
//
//   XS(s,t)   R_st = R_ss^-T X_st by forward substitution over the 16-row panels of
//             R_ss, which F(s) publishes one by one while it is still factoring
//             (xs_run follows its counter): replaces P(s,t), needs no W_ss
//   XSF(s+1)  XS(s,s+1), then A_s+1,s+1 -= R_s,s+1^T R_s,s+1 with R_s,s+1 still in LDS
//             (replaces S(s,s+1,s+1)), then F(s+1) on the tile where it lies. These
//             are the spine tasks: three dedicated workgroups take them round robin,
//             XSF(s+1) solving beside the leaf phase of XSF(s).
//
// Round 5 splits the spine and makes its hand-offs data (default; GPX_PANEL_SPLIT=0 /
// GPX_PANEL_FOLD=0 give the round-2 graph back, lock-step sweeps keep the fused task):
//
//   XS(s,t)   every row-panel solve of the rows below the first FOLDS the last update of its
//             tile in itself: X_st -= R_{s-1,s}^T R_{s-1,t}, rank-16 update by rank-16 update,
//             following the rows of the two solves above it while they are being produced
//             (the S(s-1,s,t) products leave the graph). XS(s,s+1) and XS(s,s+2) run on
//             the spine's workgroups, the others in the general queue.
//   UF(s+1)   a spine task on a workgroup of its own: A_s+1,s+1 -= R_s,s+1^T R_s,s+1 FOLLOWING
//             the rows of XS(s,s+1) as they are stored, then F(s+1) from LDS (uf_run).
//   hand-offs the leaf's panels go to the tile's MAILBOX (the diagonal tile of the staging
//             matrix) as well, and the row-panel tiles start out filled with a pattern no
//             computation produces (panel_sentinel_kernel, in front of the launch): whoever
//             follows polls the DATA -- a chunk that still shows the pattern is asked for
//             again -- instead of a counter behind a drain of the producer's stores. 16-B sc1
//             stores and loads are not torn, so a chunk is either old or new. The counters
//             remain for the claim-time dependencies and for strict mode.
//
// Per tile of the chain (N = 2048): 42.6 us (round 4) -> 39.6 (counters polled together, panels
// requested two steps ahead) -> 31 (split + fold + data hand-offs): leaf 19.6, the next tile's
// solve ends 5.3 us behind it, its follower 3.7 us behind that, D - X^T X and the way to LDS
// 1.8. profiles/r05_panel_whole_2048_traces.txt.
//
// Queue discipline: the host orders the tasks by a list-scheduling simulation
// (critical path first), which is a topological order; a workgroup takes the
// next index with one atomic and then waits for that task's counters. Every
// task a workgroup can be waiting for has a smaller index, so it has already
// been claimed by a workgroup that is running: the smallest unfinished task
// never waits, and the kernel drains whatever the residency of the grid is.
// Waits are bounded by wall-clock time as well; a timeout raises the abort flag,
// every workgroup leaves, and the host reports an error instead of hanging.
// The last workgroup out clears the control block for the next launch.
//
// Replaces the diagonal-block part of LAPACK dpotrf / dtrtri reached from
// /root/reference/pygp/inference/exact.py:54,129.

#include "gemm_tile.h"
#include "leaf_dev.h"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>

#define PT_LEAF 0
#define PT_GEMM_TN 1        // op(A)[m][k] = A[k][m], B[k][n]
#define PT_GEMM_NN 2        // op(A)[m][k] = A[m][k], B[k][n]
#define PT_XS 3             // row-panel tile solved alongside the leaf of its row (xs_run)
#define PT_UF 4             // round 5: diagonal update of tile (t,t) FOLLOWING the solve of tile
                            // (t-1,t) on another workgroup, then the leaf of that tile (uf_run)
// 16-B chunks of a tile that its follower has not seen yet hold this pattern (a signalling
// NaN no computation produces); written by panel_sentinel_kernel in front of the launch
#define PT_SENTINEL 0x7FF4A5A55A5AA5A5ll
#define PCTL_HEAD 4         // ctl[0] next task, [1] workgroups gone, [2] abort
// the two gate counters of wide panels sit behind the counters of the largest graph
// (a whole matrix of GPX_PANEL_WHOLE_MAX: T = 32, 3 T^2 + T counters; T = E = 8 needs
// fewer) and are never cleared
#define PCTL_TMAX (GPX_PANEL_WHOLE_MAX / 128)
#define PCTL_GATES (PCTL_HEAD + (PCTL_TMAX + 1) * (PCTL_TMAX + 1) + 2 * PCTL_TMAX * PCTL_TMAX + PCTL_TMAX)
#define PANEL_IG 8          // tiles per inverse group: W is assembled inside the 1024-blocks
                            // of the blocked driver only, whatever the launch covers
#define SUB 64              // edge of a product task

// Task offsets are stored for a SYMBOLIC row stride of PT_LD elements (row * PT_LD + column,
// columns < PT_LD) and turned into element offsets with the launch's real stride on the
// device (pt_off): a task list then depends on the shape of the graph alone, not on the
// leading dimension of the matrix it runs on -- one list per (tiles, extra columns, workers)
// whatever sizes a process sweeps through.
#define PT_LD_SHIFT 20
#define PT_LD (1 << PT_LD_SHIFT)
__host__ __device__ __forceinline__ long long pt_off(long long o, int ld)
{
    return (o >> PT_LD_SHIFT) * ld + (o & (PT_LD - 1));
}

struct PTask {
    long long offA, offB, offCin, offCout;   // symbolic offsets (pt_off) inside their buffers
    int op, klo, khi, goff;
    short bufA, bufB, bufCin, bufCout;       // 0 = A (R), 1 = W, 2 = X (scratch)
    short neg, beta1, ndep, sig;
    short siginc, sub, sig2, spine;          // spine: 1 = a PT_XS task of the spine's list;
                                             // sub: edge of the product tile (64, or 32 for
                                             // the two products on the critical path);
                                             // sig2: PT_XS with the diagonal update, counter
                                             // of R_st (moved by STAGE as soon as R_st is out)
    short dep[4], thr[4];
    long long offFA, offFB;                  // fold != 0: the two row-panel tiles R(s-1,s), R(s-1,s+1)
                                             // whose product is the last update of this task's tile
    short fold, nhost;                       // nhost: dependencies beyond ndep that only the host
    short pad3[2];                           // looks at (order of the queue, checks): the tasks
                                             // whose DATA this one polls
};

struct PanelArgs {
    double *bA, *bW, *bX;                    // A (-> R), W, X (scratch): blocks' origins
    int ld;
    const PTask *tasks;                      // every task but the leaves, in schedule order
    const PTask *spine;                      // the chain of the diagonal tiles, task i on
                                             // spine workgroup i mod nspwg
    int ntasks, nspine, nspwg, nctr;
    // Member-batched launch (round 4): the SAME task graph for nmem matrices that lie
    // mstride elements apart, each with its own control block (pstride ints apart) and info
    // word. The general queue interleaves the members -- position t is task t / nmem of
    // member t % nmem, which keeps it a topological order of the union of the (independent)
    // graphs -- and every member has nspwg spine workgroups of its own (workgroup g <
    // nspwg * nmem is spine g / nmem of member g % nmem). The queue head, the count of
    // workgroups gone and the abort flag are those of member 0's block.
    int nmem;
    long long mstride;
    int pstride;
    // gates (wide panels behind the look-ahead): counters OUTSIDE the control block that
    // other streams move while this launch runs -- dependency index nctr + g waits for
    // gates[g] >= gate_need<g>. They only ever grow; the host knows how far they will
    // have been moved by the launches it has enqueued in front of the ones this panel
    // has to wait for.
    const int *gates;
    int gate_need0, gate_need1;
    int *ctl;
    int *info;
    int goff;
    long long timeout;                       // wall_clock64 ticks (100 MHz)
    int leafskip;                            // timing experiments (GPX_PANEL_LEAF_SKIP), 0
    int strict;                              // GPX_PANEL_STRICT=1: agent-scope release /
                                             // acquire fences around every hand-off
    volatile int *dbg;                       // GPX_PANEL_DEBUG: host-visible progress log
    long long *trace;                        // GPX_PANEL_DEBUG=2: [task][claim, start, end, wg]
};

// What a task body needs of the launch, by value. The bodies are inlined into the one
// kernel (as separate functions they get 256 VGPRs and spill: only kernels can ask for
// one wave per SIMD); each makes `tid` opaque at its top, see xs_run.
struct PanelCtx {
    double *bA, *bW, *bX;
    int *ctl;                                // this member's control block
    int *gctl;                               // the launch's (abort flag)
    int *info;
    long long timeout;
    int ld, goff, strict, leafskip;
};

typedef Geo<SUB, 2, 2> PG;                   // 256 threads, wave 32x32
typedef Geo<32, 2, 2> PG32;                  // 256 threads, wave 16x16: row panel and update of
                                             // the NEXT diagonal tile, 16 tasks each instead of 4
typedef Geo<128, 2, 2> PG128;                // 256 threads, wave 64x64: throughput work off the
                                             // chain (the sums of the whole inverse): half the
                                             // operand bytes per flop of a 64-tile, and a lone
                                             // workgroup is bound by the bytes it can request

// buffer id of a task -> pointer (no dynamic indexing of the kernel arguments:
// that would put them, and every local array with them, into scratch)
__device__ __forceinline__ double *panel_buf(const PanelCtx &p, int id)
{
    return id == 0 ? p.bA : (id == 1 ? p.bW : p.bX);
}

// One 64x64 product: Cout = alpha * op(A) B + beta * Cin over k in [klo, khi),
// khi - klo a multiple of 64. A lone workgroup per CU cannot hide the global-load
// latency behind other waves, so the operands are requested a GROUP (4 slices, 64
// of k) at a time and two groups are always in flight: a K <= 128 product -- the
// ones on the critical path -- issues every load before its first MFMA.
// (a 128-tile's group is two slices, 32 of k: the same 64 KB a group, 128 VGPRs for the two in flight)
template <typename G> struct SliceGroup {
    static constexpr int NS = G::TILE > 64 ? 2 : 4;
    Regs<G::NLOAD> a[NS], b[NS];
};

// Tiles change hands between workgroups (and XCDs, each with its own L2) while the
// kernel runs. Every access to them is an agent-scope (sc1) access, coherent at
// the memory side, so that a hand-off needs no L2 write-back (release) and no L2
// invalidate (acquire): with those two fences per task a 128 KB tile copy took
// 18 us.
//
// INVARIANT the fast path rests on (gfx950 behaviour, not the HIP memory model):
//   (1) EVERY load and store of a tile that another workgroup wrote or will read is an
//       sc1 access -- leaf2_run<true> (its gload / gstore lambdas) and panel_gemm
//       (agent_load16 / agent_store16 of operands, Cin and Cout). One plain access added to
//       either would read a stale L1 / L2 line or leave a dirty one behind, silently;
//   (2) every storing wave drains its stores (s_waitcnt 0) and the workgroup meets at a
//       barrier before lane 0 moves the task's counter;
//   (3) the consumer's wave 0 polls the counter with sc1 loads and the other waves are
//       released by the barrier that follows.
// GPX_PANEL_STRICT=1 brackets every hand-off with agent-scope release / acquire fences
// as the memory model asks (L2 write-back before the counter moves, L1 / L2 invalidate
// after the poll): slower, and bit-identical if the invariant holds
// (tests/test_gpu_la.py::test_panel_kernel_strict_handoffs).
template <typename G, bool KMAJOR>
__device__ __forceinline__ Regs<G::NLOAD> panel_load_slice(__amdgpu_buffer_rsrc_t P, int ld,
                                                           int k0, int tid)
{
    Regs<G::NLOAD> out;
#pragma unroll
    for (int c = 0; c < G::NLOAD; ++c) {
        const int idx = tid + G::NTH * c;
        if (KMAJOR) {
            const int row = idx / (G::TILE / 2), c2 = idx % (G::TILE / 2);
            out.v[c] = agent_load16(P, ((k0 + row) * ld + 2 * c2) * 8);
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            out.v[c] = agent_load16(P, (row * ld + k0 + 2 * k2) * 8);
        }
    }
    return out;
}

template <int TA, typename G>
__device__ __forceinline__ SliceGroup<G> load_group(__amdgpu_buffer_rsrc_t A,
                                                    __amdgpu_buffer_rsrc_t B, int ld, int k0,
                                                    int tid)
{
    SliceGroup<G> g;
#pragma unroll
    for (int s = 0; s < SliceGroup<G>::NS; ++s) {
        g.a[s] = panel_load_slice<G, TA == 1>(A, ld, k0 + s * BK, tid);
        g.b[s] = panel_load_slice<G, true>(B, ld, k0 + s * BK, tid);
    }
    return g;
}

template <int TA, typename G>
__device__ __forceinline__ void compute_group(const SliceGroup<G> &g, double *smem, int tid,
                                              const double *ap0, const double *bp0,
                                              v4d (&acc)[G::WTM][G::WTN])
{
    constexpr bool AKM = (TA == 1);
    constexpr int AK = AKM ? 4 * G::KSTR : 4, AT = AKM ? 16 : 16 * MNSTR;
    constexpr int BKS = 4 * G::KSTR, BT = 16;
    double *As = smem, *Bs = smem + 2 * G::OPER;
#pragma unroll
    for (int s = 0; s < SliceGroup<G>::NS; ++s) {
        const int buf = s & 1;
        store_slice<G, AKM>(As + buf * G::OPER, tid, g.a[s]);
        store_slice<G, true>(Bs + buf * G::OPER, tid, g.b[s]);
        __syncthreads();
        mfma_slice<G::WTM, G::WTN, AK, AT, BKS, BT>(ap0 + buf * G::OPER, bp0 + buf * G::OPER,
                                                    acc);
    }
}

template <int TA, typename G>
__device__ __forceinline__ void panel_gemm(const double *__restrict__ Ap,
                                           const double *__restrict__ Bp, int ld,
                                           const double *Cin, double *Cout, int klo, int khi,
                                           double alpha, double beta, double *smem, int tid)
{
    // operand tiles as raw buffers: 16-B sc1 loads (see leaf_dev.h)
    __amdgpu_buffer_rsrc_t A = agent_rsrc(Ap), B = agent_rsrc(Bp);
    constexpr int TS = G::TILE, WT = TS / 2;             // tile edge, wave tile edge
    constexpr int WTM = G::WTM, WTN = G::WTN;
    constexpr bool AKM = (TA == 1);
    constexpr int C2 = TS / 2;                           // double2 per row of the C tile
    constexpr int NC = TS * C2 / 256;                    // double2 per thread of the C tile
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;

    // beta * Cin enters as the initial accumulator value (alpha = +-1: exact), so its
    // loads travel with the first operand loads instead of after the last MFMA. The
    // tile comes in as 16-B sc1 loads (row-major chunks) and is dealt to the MFMA
    // accumulator layout through LDS: the 8-B per-element form runs at about half the
    // 16-B rate (leaf_dev.h)
    constexpr int CS = TS + 2;                           // row stride of the staged C tile
    // (a 128-tile's image does not fit beside the operand buffers: it takes their place,
    // before the first slice is stored and after the last one is read)
    constexpr bool ALIAS = TS > 64;
    double *Cs = ALIAS ? smem : smem + 4 * G::OPER;      // beyond the operand buffers
    v4d acc[WTM][WTN];
    if (beta != 0.0) {
        __amdgpu_buffer_rsrc_t rC = agent_rsrc(Cin);
        double2 cin[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int idx = tid + 256 * i, row = idx / C2, c2 = idx % C2;
            cin[i] = agent_load16(rC, (row * ld + 2 * c2) * 8);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int idx = tid + 256 * i, row = idx / C2, c2 = idx % C2;
            *reinterpret_cast<double2 *>(Cs + row * CS + 2 * c2) = cin[i];
        }
        __syncthreads();
        const double cscale = beta / alpha;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = cscale * Cs[(wm * WT + i * 16 + lk + 4 * r) * CS +
                                               wn * WT + j * 16 + lr];
        if (ALIAS) __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    constexpr int GK = BK * SliceGroup<G>::NS;           // k of one group
    const int ngroups = (khi - klo) / GK, last = ngroups - 1;
    double *As = smem, *Bs = smem + 2 * G::OPER;
    const int amn = wm * WT + lr, bmn = wn * WT + lr;
    const double *ap0 = As + (AKM ? lk * G::KSTR + amn : amn * MNSTR + lk);
    const double *bp0 = Bs + lk * G::KSTR + bmn;
    SliceGroup<G> g0 = load_group<TA, G>(A, B, ld, klo, tid);
    SliceGroup<G> g1 = load_group<TA, G>(A, B, ld, klo + GK * min(1, last), tid);
    // pairs of groups, then the odd one (a conditional use of g1 inside the loop
    // sends that register set to scratch)
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
        compute_group<TA, G>(g0, smem, tid, ap0, bp0, acc);
        g0 = load_group<TA, G>(A, B, ld, klo + GK * min(g + 2, last), tid);
        compute_group<TA, G>(g1, smem, tid, ap0, bp0, acc);
        g1 = load_group<TA, G>(A, B, ld, klo + GK * min(g + 3, last), tid);
    }
    if (g < ngroups) compute_group<TA, G>(g0, smem, tid, ap0, bp0, acc);

    // result out through the same LDS tile: 16-B sc1 stores (an 8-B sc1 store costs
    // 2.7x the time per byte)
    __syncthreads();                                     // Cs: the Cin reads are over
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * WT + i * 16 + lk + 4 * r) * CS + wn * WT + j * 16 + lr] =
                    alpha * acc[i][j][r];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rO = agent_rsrc(Cout);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int idx = tid + 256 * i, row = idx / C2, c2 = idx % C2;
        agent_store16(rO, (row * ld + 2 * c2) * 8,
                      *reinterpret_cast<const double2 *>(Cs + row * CS + 2 * c2));
    }
}

// ---- XS(s,t): the row-panel tile R_st, solved alongside the leaf of its row ------
//
// R_st = R_ss^-T X_st by blocked forward substitution over the 16-row panels of R_ss,
// each taken from memory the moment the leaf has published it (leaf2_run's stream
// counter): X[p] <- Y_p X[p] (Y_p = U_pp^-T, the transposed diagonal block of W_ss),
// X[q] -= R_ss[p][q]^T X[p] for q > p. The tile finishes one panel hand-off after the
// leaf's last pivot instead of leaf-inverse + tile store + hand-off + product later,
// and no longer needs W_ss: the leaf's inverse is off the critical path.
// With `syrk` (t = s+1) the same workgroup goes on to the next diagonal tile,
// A_tt -= R_st^T R_st with R_st still in LDS (upper 16-blocks), which replaces a tile
// store, a hand-off and a reload on the critical path.
// Each wave owns two 16-column strips (solve and updates of a strip are wave-local,
// one barrier pair per panel for the shared panel buffer) and prefetches panel p+1
// into registers before it works on panel p whenever the leaf is that far ahead.
#define XRS 114                              // LDS row stride of the panel buffer (doubles)

struct XsPanelRegs {
    double2 y, r[4];
};

// Panel pp of R_ss -- the 16 x (112 - 16 pp) doubles right of its diagonal 16-block and that
// block's inverse -- from memory to registers and from there to the LDS panel buffer. Since
// round 5 every thread issues a FIXED number of loads per panel (slots enumerated densely
// over the 256 threads, idle threads repeat the last slot), with pp a compile-time constant
// of the unrolled solve: the loads of a panel are then a straight-line sequence and the
// s_waitcnt the compiler puts in front of the panel's commit waits for exactly that panel,
// not for everything in flight (with lane conditions around the loads it branched around
// them on an empty exec mask and had to assume the worst: vmcnt(0) at every commit).
__device__ __forceinline__ constexpr int xs_nload(int pp) { return (16 * (56 - 8 * pp) + 255) / 256; }

__device__ __forceinline__ void xs_issue(XsPanelRegs &g, __amdgpu_buffer_rsrc_t rR,
                                         __amdgpu_buffer_rsrc_t rW, int ld, int pp, int tid)
{
    const int i0 = 16 * pp, ncol2 = 56 - 8 * pp;         // double2 per row right of the block
    {
        const int e = tid & 127, r = e >> 3, c = e & 7;  // (threads 128-255 repeat 0-127)
        g.y = agent_load16(rW, ((i0 + r) * ld + i0 + 2 * c) * 8);
    }
    const int slots = 16 * ncol2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < xs_nload(pp)) {
            const int e = min(tid + 256 * j, slots - 1), r = e / ncol2, c2 = e % ncol2;
            g.r[j] = agent_load16(rR, ((i0 + r) * ld + i0 + 16 + 2 * c2) * 8);
        }
}

__device__ __forceinline__ void xs_commit(const XsPanelRegs &g, double *Rp, double *Yp, int pp,
                                          int tid)
{
    const int ncol2 = 56 - 8 * pp, slots = 16 * ncol2;
    if (tid < 128) {
        const int r = tid >> 3, c = tid & 7;             // W[r][2c..]: Y[k][r]
        Yp[(2 * c) * YS + r] = g.y.x;
        Yp[(2 * c + 1) * YS + r] = g.y.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < xs_nload(pp)) {
            const int e = tid + 256 * j, r = e / ncol2, c2 = e % ncol2;
            if (e < slots) *reinterpret_cast<double2 *>(Rp + r * XRS + 2 * c2) = g.r[j];
        }
}

// upper 16-block b (row-major over q <= r) -> q, r. (Tables: as constexpr functions with
// loops they were not folded and ran as 72 scalar loops in front of the product.)
__device__ static constexpr unsigned char XS_BLOCK_Q[36] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};
__device__ static constexpr unsigned char XS_BLOCK_R[36] = {0, 1, 2, 3, 4, 5, 6, 7, 1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};

// X^T X on the 36 upper 16-blocks of the tile in X, nine per wave: wave W takes block rows
// W (8 - W blocks) and 7 - W (W + 1 blocks). One instantiation per wave, so that a k-step
// reads each of its 8 - W operand columns ONCE into registers (runtime column offsets
// needed 18 LDS reads per 9 MFMAs and the four waves kept the LDS half busy: 93 cycles
// per MFMA); everything from the birth of the accumulators (in the MFMAs of k = 0:
// initialised with moves and carried through the loop they were kept in VGPRs and copied
// to AGPRs and back around every MFMA) to their store lives inside the instantiation,
// so that no accumulator crosses the switch. k outermost, nine independent
// accumulators, the operands of step k + 1 read while the MFMAs of step k run.
// R_st has gone out during the solve (xs_run), its last 16 rows and the loads of the old
// diagonal tile D just before this product; the early signal follows a few k-steps in.
template <int W, class F>
__device__ __forceinline__ void xs_syrk(double *__restrict__ X, int tid, int *sig2, int strict,
                                        long long *tr, F &&store)
{
    const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int QA = W, QB = NBK - 1 - W;              // the two block rows
    const double *xrow = X + lk * LS + lr;
    double x0[NBK], x1[NBK];                             // columns W..7 of one k-step
    v4d acc[9];
    auto operands = [&](double (&x)[NBK], int k) {
        const double *row = xrow + 4 * k * LS;
#pragma unroll
        for (int c = W; c < NBK; ++c) x[c] = row[16 * c];
    };
    auto products = [&](const double (&x)[NBK], bool first) {
        int j = 0;
#pragma unroll
        for (int r = QA; r < NBK; ++r, ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                x[QA], x[r], first ? (v4d){0.0, 0.0, 0.0, 0.0} : acc[j], 0, 0, 0);
#pragma unroll
        for (int r = QB; r < NBK; ++r, ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                x[QB], x[r], first ? (v4d){0.0, 0.0, 0.0, 0.0} : acc[j], 0, 0, 0);
    };
    operands(x0, 0);
    operands(x1, 1);
    products(x0, true);
    store(0);
    operands(x0, 2);
    products(x1, false);
    store(1);
#pragma unroll 1
    for (int k = 2; k < 32; k += 2) {
        if (k == 10) {
            // the last 32 rows of R_st go out under the first eight k-steps (store(j), one
            // 16-B store per thread each: issued back to back in front of the product they
            // held the wave for 1.1 us per 16 rows while the store path of the CU drained),
            // the others went during the solve: two k-steps later they are at the memory
            // side and the updates that read R_st, the tiles of the next row panel first,
            // may start
            if (tr && tid == 0) tr[9] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads of D are older
            __syncthreads();
            if (tid == 0 && sig2) {                       // (no counter: lock-step sweeps)
                if (strict) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add(sig2, 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        operands(x1, k + 1);
        products(x0, false);
        if (k < 8) store(k);
        operands(x0, min(k + 2, 31));
        products(x1, false);
        if (k < 8) store(k + 1);
    }
    if (tr && tid == 0) tr[10] = wall_clock64();
    __syncthreads();                                     // nobody reads X any more
    int j = 0;
#pragma unroll
    for (int r = QA; r < NBK; ++r, ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) X[(16 * QA + lk + 4 * t) * LS + 16 * r + lr] = acc[j][t];
#pragma unroll
    for (int r = QB; r < NBK; ++r, ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) X[(16 * QB + lk + 4 * t) * LS + 16 * r + lr] = acc[j][t];
}

// returns false when the wait for the leaf timed out / the launch is being aborted
// PRE (lock-step sweeps with dense row panels, round 5): the tile has been updated AND solved by
// sweep_xs_kernel already -- R_st comes in from A and the task goes straight to the diagonal
// update and the leaf (the same code from there on: the same bits)
template <bool PRE = false>
__device__ __forceinline__ bool xs_run(PanelCtx p, const PTask *tkp, long long *tr)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const PTask &tk = *tkp;
    // (tid is made opaque in every task body: everything derived from it is loop-invariant
    // in the kernel's task loop, got hoisted to the top of the kernel and spilled there --
    // 49 VGPRs reloaded from scratch inside the bodies, 2.5-4 us per task)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const long long t0 = wall_clock64();
    double *X = reinterpret_cast<double *>(smem_raw);    // [128][LS]
    double *Rp = X + LB * LS;                            // [16][XRS] R_ss[p][p+1..]
    double *Yp = Rp + 16 * XRS;                          // [16][YS]
    int *flag = reinterpret_cast<int *>(Yp + 16 * YS);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int ld = p.ld;
    // bufA == 2: the panels of R_ss come from the tile's mailbox (leaf2_run), polled as data;
    // otherwise from R and W themselves (lock-step sweeps: they are final there)
    const bool mail = __builtin_amdgcn_readfirstlane((int)tk.bufA) == 2;
    __amdgpu_buffer_rsrc_t rR = agent_rsrc((mail ? p.bX : p.bA) + pt_off(tk.offA, ld)),
                           rW = agent_rsrc((mail ? p.bX : p.bW) + pt_off(tk.offA, ld));
    __amdgpu_buffer_rsrc_t rX = agent_rsrc(PRE ? p.bA + pt_off(tk.offCout, ld)
                                               : p.bX + pt_off(tk.offB, ld)),
                           rO = agent_rsrc(p.bA + pt_off(tk.offCout, ld));
    int *ctl = p.ctl;
    const int *cy = ctl + PCTL_HEAD + __builtin_amdgcn_readfirstlane(tk.klo);

    if (tid == 0) flag[0] = 0;
    {   // tile in
        double2 tmp[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e2 = tid + 256 * i;
            tmp[i] = agent_load16(rX, ((e2 >> 6) * ld + 2 * (e2 & 63)) * 8);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e2 = tid + 256 * i;
            *reinterpret_cast<double2 *>(X + (e2 >> 6) * LS + 2 * (e2 & 63)) = tmp[i];
        }
    }
    __syncthreads();
    if constexpr (!PRE) {
    // this wave's two 16-column strips live in registers from here on (MFMA accumulator
    // layout: xr[cc][q][r] = X[16q + lk + 4r][c0 + lr]); with the strips in LDS every
    // update waited for its accumulator reads and a panel took 4.7 us instead of 2
    v4d xr[2][NBK];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int q = 0; q < NBK; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xr[cc][q][r] = X[(16 * q + lk + 4 * r) * LS + 16 * (wave + 4 * cc) + lr];

    // The last update of the task's own tile, folded in HERE (round 5, fold). This task solves
    // tile (s, t); its tile still lacks X -= R(s-1,s)^T R(s-1,t), and both operands are being
    // produced right now, row block by row block, by two solves of tile row s-1 that run
    // beside the leaf of tile s-1. Until round 4 worker products applied that update after
    // both solves had finished -- signal, claim, operand loads, product, store, signal, this
    // task's poll and then its 5-us tile-in: 13-16 us between "R(s-1,s) is complete" and this
    // task's first solve step, which then ran BEHIND its leaf at its own pace (17 us) and
    // ended 10 us after the leaf did; and the same chain ran down every tile column, so that
    // the columns two and three right of the diagonal held the diagonal back once the spine
    // itself was faster. Now the task holds its tile in registers and follows the rows of the
    // two producers -- polled as data, like everything else that is handed over along the
    // chain: rank-16 update by rank-16 update (64 MFMAs a wave, 1.8 us, against 2.4 us per row
    // block of the producers; as a product task the same update cost 40 us of workgroup
    // time). k ascends in the groups of four of the product tasks and -(a) b accumulates into
    // X itself: their bits.
    if (__builtin_amdgcn_readfirstlane((int)tk.fold)) {
        __amdgpu_buffer_rsrc_t rFA = agent_rsrc(p.bA + pt_off(tk.offFA, ld)),
                               rFB = agent_rsrc(p.bA + pt_off(tk.offFB, ld));
        __syncthreads();                                 // every wave has its strips out of X
        if (p.strict) {
            // strict mode: wait for the producers' end signals (the release the acquire pairs with)
            const int nd = __builtin_amdgcn_readfirstlane((int)tk.ndep);
            for (int i = 0; i < 2; ++i) {
                const int *cs = ctl + PCTL_HEAD + __builtin_amdgcn_readfirstlane((int)tk.dep[nd + i]);
                const int need = __builtin_amdgcn_readfirstlane((int)tk.thr[nd + i]);
                for (;;) {
                    const int hv = __hip_atomic_load(cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_readfirstlane(hv) >= need) break;
                    if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                        if (lane == 0) {
                            __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            flag[0] = 1;
                        }
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        // 16 rows x 128 columns of each operand per step: four 16-B loads per thread and
        // operand (this wave: rows wave + 4 i), asked for one step ahead, staged in the (free)
        // tile buffer: rows [0, 16) and [16, 32) of X, the next step in rows [32, 64)
        double2 fa[2][4], fb[2][4];
        auto issue_f = [&](double2 (&a)[4], double2 (&b)[4], int j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = ((16 * j + wave + 4 * i) * ld + 2 * lane) * 8;
                a[i] = agent_load16(rFA, off);
                b[i] = agent_load16(rFB, off);
            }
        };
        auto fresh_f = [&](const double2 (&a)[4], const double2 (&b)[4]) -> bool {
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bad = bad || __double_as_longlong(a[i].x) == PT_SENTINEL ||
                      __double_as_longlong(a[i].y) == PT_SENTINEL ||
                      __double_as_longlong(b[i].x) == PT_SENTINEL ||
                      __double_as_longlong(b[i].y) == PT_SENTINEL;
            return __ballot(bad) == 0ull;
        };
        issue_f(fa[0], fb[0], 0);
        bool deadf = false;
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            if (deadf) continue;
            while (!fresh_f(fa[j & 1], fb[j & 1])) {
                const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                issue_f(fa[j & 1], fb[j & 1], j);
                if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                    if (lane == 0) {
                        __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 1;
                    }
                    break;
                }
            }
            double *stA = X + 32 * (j & 1) * LS, *stB = stA + 16 * LS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<double2 *>(stA + (wave + 4 * i) * LS + 2 * lane) = fa[j & 1][i];
                *reinterpret_cast<double2 *>(stB + (wave + 4 * i) * LS + 2 * lane) = fb[j & 1][i];
            }
            if (j + 1 < NBK) issue_f(fa[(j + 1) & 1], fb[(j + 1) & 1], j + 1);
            __syncthreads();                             // (one barrier a step: the buffers alternate)
            if (__builtin_amdgcn_readfirstlane(flag[0])) {
                deadf = true;
                continue;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double *ra = stA + (4 * ks + lk) * LS + lr;
                const double *rb = stB + (4 * ks + lk) * LS + 16 * wave + lr;
                double a[NBK];
#pragma unroll
                for (int q = 0; q < NBK; ++q) a[q] = -ra[16 * q];
                const double b0 = rb[0], b1 = rb[64];
#pragma unroll
                for (int q = 0; q < NBK; ++q) {
                    xr[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b0, xr[0][q], 0, 0, 0);
                    xr[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1, xr[1][q], 0, 0, 0);
                }
            }
        }
        if (deadf) return false;
        __syncthreads();                                 // the staging rows are free again
        if (tr && tid == 0) tr[14] = wall_clock64();
    }

    // Row block p of X = R_st is final after its solve in step p and goes out to memory in
    // step p + 1 (four 16-B stores per thread; every wave wrote its strips to LDS before
    // the barriers of that step). A lone CU stores ~23 GB/s and a wave cannot issue past
    // its queued stores: the whole tile at the end was 5.7 us between the last solve and
    // "R_st is in memory", which the updates of the next row panel wait for.
    auto rows_out = [&](int pb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e2 = tid + 256 * i;
            const int r = 16 * pb + (e2 >> 6), c = 2 * (e2 & 63);
            agent_store16(rO, (r * ld + c) * 8,
                          *reinterpret_cast<const double2 *>(X + r * LS + c));
        }
    };

    // The leaf's counter: three signals per published panel. Panels are requested TWO steps
    // ahead of their use, in a fixed pattern -- the request for panel pp + 2 at the top of
    // step pp waits for that panel's publication first -- and the eight steps are unrolled:
    // straight-line loads, so that a commit waits for its own panel only (see xs_issue). A
    // solve that runs beside its leaf ends where it ended before (its last steps still start
    // with the leaf's last panel); one that starts late, or whose leaf has long finished (the
    // row-panel tasks of a lock-step sweep, most worker tasks of a panel launch), no longer
    // stalls for a memory round trip per step: rounds 2-4 requested one panel ahead inside a
    // rolled loop and every commit waited for everything in flight (2.3 us a step whatever
    // was there; the MFMAs of a step are 0.25-1.8 us).
    XsPanelRegs g[3];
    int have = 0;
    auto wait_pub = [&](int n) {                         // panels [0, n) are published
        while (have < n) {
            const int hv = __hip_atomic_load(cy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
            have = __builtin_amdgcn_readfirstlane(hv) / 3;
            if (have >= n) break;
            if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                if (lane == 0) {
                    __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 1;
                }
                have = NBK;                              // fall through; the flag ends the task
            }
        }
        if (p.strict) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    // Round 5, second half: with a mailbox the panels are asked for WITHOUT waiting for their
    // publication -- a request that comes too early returns the pattern and is repeated at the
    // commit until the data is there (the data is its own flag: a chunk is seen one
    // store-to-load latency after the leaf stored it, where the counter needed the leaf's
    // drain one step later, its atomic and this task's poll -- about 4 us a panel, and with
    // it the leaf's last panel on the chain). The counter is still followed in strict mode
    // (the memory model's acquire needs something to pair with) and without a mailbox.
    const bool usectr = !mail || p.strict;
    auto fresh = [&](const XsPanelRegs &v, int pp) -> bool {
        bool bad = __double_as_longlong(v.y.x) == PT_SENTINEL ||
                   __double_as_longlong(v.y.y) == PT_SENTINEL;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < xs_nload(pp))
                bad = bad || __double_as_longlong(v.r[j].x) == PT_SENTINEL ||
                      __double_as_longlong(v.r[j].y) == PT_SENTINEL;
        return __ballot(bad) == 0ull;
    };
    if (usectr) wait_pub(1);
    xs_issue(g[0], rR, rW, ld, 0, tid);
    if (usectr) wait_pub(2);
    xs_issue(g[1], rR, rW, ld, 1, tid);
    bool dead = false;
#pragma unroll
    for (int pp = 0; pp < NBK; ++pp) {
        if (dead) continue;                              // (wave- and workgroup-uniform)
        // (trace: step stamps of a task that runs no leaf afterwards -- the leaf's slots)
        if (tr && tid == 0 && !tk.beta1) tr[16 + pp] = wall_clock64();
        if (pp + 2 < NBK) {
            if (usectr) wait_pub(pp + 3);
            xs_issue(g[(pp + 2) % 3], rR, rW, ld, pp + 2, tid);
        }
        if (mail) {
            // this wave's part of panel pp: asked for again until it is there. One round trip
            // a turn and nothing else in it: the abort flag rides along as one more load of the
            // same turn (lane-uniform address, the value is read after the panel's)
            while (!fresh(g[pp % 3], pp)) {
                const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                xs_issue(g[pp % 3], rR, rW, ld, pp, tid);
                if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                    if (lane == 0) {
                        __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 1;
                    }
                    break;
                }
            }
        }
        __syncthreads();                                 // panel pp-1 has been used by all
        xs_commit(g[pp % 3], Rp, Yp, pp, tid);
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(flag[0])) {
            dead = true;
            continue;
        }
        // Rows of the previous step out, BEHIND those loads: the stores of a row block take
        // the CU's store path ~1 us to drain, the matrix pipe is not held by them, but the
        // next vector-memory instruction is. (The rows of the last two steps go out under the
        // diagonal update, or at the end.)
        // (a task without the diagonal update sends rows 6 in step 7 too: the follower of a
        // spine's solve then has only one row block left when the solve ends)
        if (pp >= 1 && (pp < NBK - 1 || !tk.beta1)) rows_out(pp - 1);

        // X[p] <- Y_p X[p]. A row of an accumulator block is k = lk + 4r when the block is
        // used as the B operand of k-step r, so the A operand takes the same k
        v4d xn[2];
        {
            double ya[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ya[r] = Yp[lr * YS + lk + 4 * r];   // Y[i][k]
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                v4d t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[r], xr[cc][pp][r], t, 0, 0, 0);
                xr[cc][pp] = t;
                xn[cc] = t;
                // the rows are final: into LDS, from where they go out to R_st
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    X[(16 * pp + lk + 4 * r) * LS + 16 * (wave + 4 * cc) + lr] = t[r];
            }
        }
        // X[q] -= R[p][q]^T X[p], q > p
#pragma unroll
        for (int q = pp + 1; q < NBK; ++q) {
            const double *rq = Rp + 16 * (q - pp - 1) + lr;
            double ra[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ra[r] = -rq[(lk + 4 * r) * XRS];   // -R[p][q][k][i]
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xr[cc][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[r], xn[cc][r], xr[cc][q],
                                                                     0, 0, 0);
        }
    }
    if (dead) return false;
    __syncthreads();                                     // X = R_st, complete
    if (tr && tid == 0) tr[4] = wall_clock64();
    if (!tk.beta1) {
        rows_out(NBK - 1);
        return true;
    }

    }

    // next diagonal tile: D -= X^T X on its upper 16-blocks
    __amdgpu_buffer_rsrc_t rD = agent_rsrc(p.bA + pt_off(tk.offCin, ld));
    // (the 36 upper 16-blocks only: 18 16-B chunks per thread; chunk e of block b is row
    // (e / 8) % 16, columns 2 (e % 8) of the block)
    double2 dv[18];
    int doff[18];                                        // row << 16 | column inside the tile
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        // block 2i or 2i + 1 of the row-major enumeration of the upper blocks
        const int e = tid + 256 * i, odd = (tid >> 7) & 1;
        const int q = odd ? XS_BLOCK_Q[2 * i + 1] : XS_BLOCK_Q[2 * i];
        const int r = odd ? XS_BLOCK_R[2 * i + 1] : XS_BLOCK_R[2 * i];
        doff[i] = ((16 * q + ((e >> 3) & 15)) << 16) | (16 * r + 2 * (e & 7));
    }
    // the old tile comes in under the product
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 18; ++i)
        dv[i] = agent_load16(rD, ((doff[i] >> 16) * ld + (doff[i] & 65535)) * 8);
    asm volatile("" ::: "memory");
    if (tr && tid == 0) tr[8] = wall_clock64();
    int *sig2 = tk.sig2 >= 0 ? ctl + PCTL_HEAD + tk.sig2 : nullptr;
    auto store = [&](int j) {                            // store j of rows_out(6), rows_out(7)
        if constexpr (PRE) return;                       // (R_st is in memory already)
        const int e2 = tid + 256 * (j & 3);
        const int r = 16 * (NBK - 2 + (j >> 2)) + (e2 >> 6), c = 2 * (e2 & 63);
        agent_store16(rO, (r * ld + c) * 8, *reinterpret_cast<const double2 *>(X + r * LS + c));
    };
    switch (wave) {
    case 0: xs_syrk<0>(X, tid, sig2, p.strict, tr, store); break;
    case 1: xs_syrk<1>(X, tid, sig2, p.strict, tr, store); break;
    case 2: xs_syrk<2>(X, tid, sig2, p.strict, tr, store); break;
    default: xs_syrk<3>(X, tid, sig2, p.strict, tr, store); break;
    }
    __syncthreads();
    if (tr && tid == 0) tr[11] = wall_clock64();
    // beta1 == 2: the tile stays in LDS for the leaf that follows in this workgroup (no
    // tile store, hand-off and reload between the last update of a diagonal tile and its
    // factorisation); otherwise it goes back to memory
    const bool keep = tk.beta1 == 2;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int r = doff[i] >> 16, c = doff[i] & 65535;
        double2 *xp = reinterpret_cast<double2 *>(X + r * LS + c);
        const double2 d = make_double2(dv[i].x - xp->x, dv[i].y - xp->y);
        if (keep) *xp = d;
        else agent_store16(rD, (r * ld + c) * 8, d);
    }
    if (keep) __syncthreads();
    if (tr && tid == 0) tr[5] = wall_clock64();
    return true;
}

// ---- UF(t): the diagonal update of tile (t,t) on a workgroup of its own (round 5) -------
//
// Until round 4 the spine task of tile t solved R_{t-1,t}, formed D -= R^T R with R still in
// its LDS and factored the tile -- 288 MFMAs a wave (8 us at the rate of one CU, 13.3 us with
// stores, signal and barriers) between the end of one leaf's solve and the start of the next
// leaf, on the chain of diagonal tiles. The product cannot hide inside the solve (another
// 8 us of MFMA on the same CU beside a 19.5-us leaf: measured three times, EXPERIMENTS.md),
// so it gets a CU of its own: this task holds the nine accumulators a wave owns, follows the
// rows of R_{t-1,t} as the solving workgroup stores them -- 16 rows at a time, through a
// small staging buffer in LDS -- and is one row block (36 MFMAs a wave) behind when the solve
// ends; then D - (the sum) and the leaf, from LDS, as before. Same k order per
// accumulator, born from zero: the bits of rounds 2-4 and of the lock-step sweep, which keeps
// the fused task.
//
// The hand-off carries no counter. The tile's 16-B chunks hold PT_SENTINEL until the solve's
// stores land (panel_sentinel_kernel fills the tile in front of the launch; sc1 stores and
// loads of 16 B are not torn): the follower polls the DATA, a row block is in when none of
// its chunks shows the pattern. The producer pays nothing -- no drain, no barrier, no
// atomic per row block (publication through counters cost the solve 0.5 us a step when it
// was tried) -- and the consumer sees a row block one store-to-load latency after it left.
struct UfCols {
    const double *c[5];                                  // block columns b .. b + 4 (mod 8)
    const double *a9, *b9;                               // the pair at distance 4
};

// One body for the four waves: X^T X is symmetric, a block may be computed as (q, r) or as
// (r, q)^T -- the same products in the same order, the same bits -- and the 36 unordered pairs
// {q, r} of block columns split into four congruent sets: with b = 2 w, wave w takes
// (b, b + d) and (b + 1, b + 1 + d), d = 0 .. 3, columns mod 8, and one of the four pairs at
// distance 4: (0,4), (2,6), (5,1), (7,3). A block whose second column wrapped around is
// written out transposed.
__device__ __forceinline__ UfCols uf_cols(const double *St, int wave, int lr, int lk)
{
    UfCols s;
    const double *row0 = St + lk * LS + lr;
    const int b = 2 * wave, s9 = b + (wave >> 1);        // 0, 2, 5, 7
#pragma unroll
    for (int c = 0; c < 5; ++c) s.c[c] = row0 + 16 * ((b + c) & 7);
    s.a9 = row0 + 16 * (s9 & 7);
    s.b9 = row0 + 16 * ((s9 + 4) & 7);
    return s;
}

// the four k-steps of one staged row block; first: the accumulators are born here
__device__ __forceinline__ void uf_ksteps(const UfCols &s, bool first, v4d (&acc)[9])
{
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        double x[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) x[c] = s.c[c][4 * n * LS];
        const double xa = s.a9[4 * n * LS], xb = s.b9[4 * n * LS];
        const bool z = first && n == 0;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            acc[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                x[0], x[d], z ? (v4d){0.0, 0.0, 0.0, 0.0} : acc[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 4; ++d)
            acc[4 + d] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                x[1], x[1 + d], z ? (v4d){0.0, 0.0, 0.0, 0.0} : acc[4 + d], 0, 0, 0);
        acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(
            xa, xb, z ? (v4d){0.0, 0.0, 0.0, 0.0} : acc[8], 0, 0, 0);
    }
}

// the nine blocks into the tile in X (upper 16-blocks; a wrapped pair as its transpose)
__device__ __forceinline__ void uf_store(double *X, int wave, int lr, int lk, const v4d (&acc)[9])
{
    const int b = 2 * wave, s9 = b + (wave >> 1);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int q0 = j < 4 ? b : (j < 8 ? b + 1 : s9), r0 = q0 + (j < 8 ? (j & 3) : 4);
        const int q = q0 & 7, r = r0 & 7;                // q0 <= 7 always
        const bool wrapped = r0 > 7;                     // then r < q: block (r, q) = this one^T
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int direct = (16 * q + lk + 4 * t) * LS + 16 * r + lr;
            const int transp = (16 * r + lr) * LS + 16 * q + lk + 4 * t;
            X[wrapped ? transp : direct] = acc[j][t];
        }
    }
}

// returns false when the rows did not arrive within the wait bound / the launch is aborted
//
// Two updates are folded in (round 5, last step). UF(t) used to wait, when it was claimed, for
// the diagonal tile with every update before step t-1 -- the last of them, step t-2, a worker's
// product of R_{t-2,t}, which at 24-32 tiles arrived 10 us after the follower should have
// started (workers are claimed in queue order and were busy). Now the task starts with the
// tile at update t-3 and follows TWO producers: first the rows of R_{t-2,t} (the second spine
// solve of tile row t-2; offFA), as the product task did it -- the accumulators start from -D
// in the MFMA layout and the tile after that step is their negative --, then the rows of
// R_{t-1,t} from zero as before; D2 = (-acc1) - acc2 element by element in registers. The
// bits of the product task followed by the fused task.
__device__ __forceinline__ bool uf_run(PanelCtx p, const PTask *tkp, long long *tr)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const PTask &tk = *tkp;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const long long t0 = wall_clock64();
    double *X = reinterpret_cast<double *>(smem_raw);    // [128][LS]; staging: rows 0..31
    int *flag = reinterpret_cast<int *>(X + LB * LS + 16 * XRS + 16 * YS);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int ld = p.ld;
    __amdgpu_buffer_rsrc_t rR = agent_rsrc(p.bA + pt_off(tk.offA, ld));      // R_{t-1,t}
    __amdgpu_buffer_rsrc_t rD = agent_rsrc(p.bA + pt_off(tk.offCin, ld));    // the old tile D
    const bool two = __builtin_amdgcn_readfirstlane((int)tk.fold) != 0;      // + step t-2
    __amdgpu_buffer_rsrc_t rR0 = agent_rsrc(p.bA + pt_off(two ? tk.offFA : tk.offA, ld));
    if (tid == 0) flag[0] = 0;

    // the old diagonal tile in the accumulator layout, negated: acc1[j][t] = -D[row][col] of
    // this wave's block j (uf_store's positions; a wrapped pair reads its mirror image in the
    // upper triangle). 8-B loads, once, long before they are needed.
    v4d acc1[9], acc2[9];
    {
        const int b = 2 * wave, s9 = b + (wave >> 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int q0 = j < 4 ? b : (j < 8 ? b + 1 : s9), r0 = q0 + (j < 8 ? (j & 3) : 4);
            const int q = q0 & 7, r = r0 & 7;
            const bool wrapped = r0 > 7;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = wrapped ? 16 * r + lr : 16 * q + lk + 4 * t;
                const int col = wrapped ? 16 * q + lk + 4 * t : 16 * r + lr;
                const double d = __builtin_bit_cast(
                    double, __builtin_amdgcn_raw_buffer_load_b64(rD, (row * ld + col) * 8, 0,
                                                                 16 /* sc1 */));
                acc1[j][t] = -d;
            }
        }
    }
    __syncthreads();                                     // flag
    if (p.strict) {
        // strict mode: the memory model's acquire needs a release to pair with -- the producers'
        // end signals (all of their tile out), not their rows one by one
        const int nd = __builtin_amdgcn_readfirstlane((int)tk.ndep);
        const int nh = __builtin_amdgcn_readfirstlane((int)tk.nhost);
        for (int i = 0; i < nh; ++i) {
            const int *cs = p.ctl + PCTL_HEAD + __builtin_amdgcn_readfirstlane((int)tk.dep[nd + i]);
            const int need = __builtin_amdgcn_readfirstlane((int)tk.thr[nd + i]);
            for (;;) {
                const int hv = __hip_atomic_load(cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(hv) >= need) break;
                if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                    if (lane == 0) {
                        __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 1;
                    }
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }

    // row block j of a tile: rows 16 j + wave + 4 i (i = 0..3) of this wave, all 128 columns
    double2 rv[2][4];
    auto issue = [&](__amdgpu_buffer_rsrc_t rT, double2 (&v)[4], int j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = agent_load16(rT, ((16 * j + wave + 4 * i) * ld + 2 * lane) * 8);
    };
    auto fresh = [&](const double2 (&v)[4]) -> bool {    // (wave-uniform) none of the pattern
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bad = bad || __double_as_longlong(v[i].x) == PT_SENTINEL ||
                  __double_as_longlong(v[i].y) == PT_SENTINEL;
        return __ballot(bad) == 0ull;
    };
    bool dead = false;
    // phase 0 (two): the rows of R_{t-2,t} into acc1 = -D; phase 1: the rows of R_{t-1,t} into
    // acc2 from zero. 16 staging steps, the buffers alternate: one barrier a step.
    issue(two ? rR0 : rR, rv[0], 0);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph == 0 && !two) continue;
        if (ph == 1 && tr && tid == 0) tr[14] = wall_clock64();
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            if (dead) continue;
            const int step = 8 * ph + j;
            __amdgpu_buffer_rsrc_t rT = ph == 0 ? rR0 : rR;
            // this wave's rows of block j: in, or asked for again until they are
            // (one round trip a turn: the abort flag rides along with the rows' loads)
            while (!fresh(rv[step & 1])) {
                const int sv = __hip_atomic_load(&p.gctl[2], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                issue(rT, rv[step & 1], j);
                if (__builtin_amdgcn_readfirstlane(sv) != 0 || wall_clock64() - t0 > p.timeout) {
                    if (lane == 0) {
                        __hip_atomic_store(&p.gctl[2], 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 1;
                    }
                    break;
                }
            }
            double *St = X + 16 * (step & 1) * LS;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<double2 *>(St + (wave + 4 * i) * LS + 2 * lane) = rv[step & 1][i];
            // the next block (of this tile, or the first of the second): may be early, checked
            // at its own step
            if (j + 1 < NBK) issue(rT, rv[(step + 1) & 1], j + 1);
            else if (ph == 0) issue(rR, rv[(step + 1) & 1], 0);
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(flag[0])) {
                dead = true;
                continue;
            }
            if (ph == 0) uf_ksteps(uf_cols(St, wave, lr, lk), false, acc1);
            else uf_ksteps(uf_cols(St, wave, lr, lk), j == 0, acc2);
        }
    }
    if (dead) return false;
    if (tr && tid == 0) tr[4] = wall_clock64();          // ("strips done": the last rows are in)
    __syncthreads();                                     // nobody reads the staging rows any more
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[j][t] = -acc1[j][t] - acc2[j][t];
    uf_store(X, wave, lr, lk, acc2);
    __syncthreads();
    if (tr && tid == 0) tr[5] = wall_clock64();
    return true;
}

// the tiles that are polled as data hold the pattern until their data lands (one workgroup a
// tile, blockIdx.x = s * TW + t): s == t < T: the mailbox of diagonal tile s (tile (s, s) of the
// staging matrix); s < t (only with `rows`): tile (s, t) of A, whose rows the solves of tile
// row s+1 and the follower of tile s+1 read while the solve of (s, t) is still producing them
__global__ __launch_bounds__(256) void panel_sentinel_kernel(double *bA, double *bX, int T, int TW,
                                                             int rows, int ld, long long mstride)
{
    const int s = (int)blockIdx.x / TW, t = (int)blockIdx.x % TW;
    if (s >= T || t < s || (t == s ? false : !rows)) return;
    double *tile = (t == s ? bX : bA) + (long long)(128 * s) * ld + 128 * t +
                   (long long)blockIdx.y * mstride;
    // (four workgroups a tile, 32 rows each: one workgroup stores ~25 GB/s)
    const double sv = __longlong_as_double(PT_SENTINEL);
    tile += (long long)(32 * blockIdx.z) * ld;
    for (int e2 = threadIdx.x; e2 < 32 * 64; e2 += 256)
        *reinterpret_cast<double2 *>(tile + (long long)(e2 >> 6) * ld + 2 * (e2 & 63)) =
            make_double2(sv, sv);
}

__device__ __forceinline__ void run_leaf(PanelCtx p, long long o, int goff, int cy,
                                                   bool fused, long long *tr)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // (the mailbox of the tile's panels: the diagonal tile of the staging matrix, unused
    // otherwise -- a launch that streams fills it with the pattern first)
    leaf2_run<true>(p.bA + o, p.ld, p.bW + o, p.ld, p.info, goff, p.leafskip, smem_raw,
                    cy >= 0 ? p.ctl + PCTL_HEAD + cy : nullptr, p.strict, fused, tr,
                    cy >= 0 ? p.bX + o : nullptr);
}

__device__ __forceinline__ void run_gemm(PanelCtx p, const PTask *tkp)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const PTask &tk = *tkp;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int ld = p.ld;
    const int op = __builtin_amdgcn_readfirstlane(tk.op);
    const double *A = panel_buf(p, tk.bufA) + pt_off(tk.offA, ld);
    const double *B = panel_buf(p, tk.bufB) + pt_off(tk.offB, ld);
    const double *Cin = panel_buf(p, tk.bufCin) + pt_off(tk.offCin, ld);
    double *Cout = panel_buf(p, tk.bufCout) + pt_off(tk.offCout, ld);
    const double alpha = tk.neg ? -1.0 : 1.0, beta = tk.beta1 ? 1.0 : 0.0;
    double *smem = reinterpret_cast<double *>(smem_raw);
    const int sub = __builtin_amdgcn_readfirstlane((int)tk.sub);
    if (sub == 128 && op == PT_GEMM_TN)
        panel_gemm<1, PG128>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
    else if (sub == 128)
        panel_gemm<0, PG128>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
    else if (op == PT_GEMM_TN && sub == 32)
        panel_gemm<1, PG32>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
    else if (sub == 32)
        panel_gemm<0, PG32>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
    else if (op == PT_GEMM_TN)
        panel_gemm<1, PG>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
    else
        panel_gemm<0, PG>(A, B, ld, Cin, Cout, tk.klo, tk.khi, alpha, beta, smem, tid);
}

__global__ __launch_bounds__(256) void panel_kernel(PanelArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_task, s_abort, s_member;
    const int tid = threadIdx.x;
    int *ctl = p.ctl;                                    // the launch's head block (member 0's)
    const int nmem = p.nmem;

    // Control flow below is kept wave-uniform on purpose (values broadcast with
    // readfirstlane, the whole of wave 0 polls): a loop whose exit the compiler
    // believes to be lane-divergent gets restructured around the workgroup
    // barriers, and waves then meet different barriers.
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup 0 is the spine: it runs the leaves and nothing else, so that a leaf
    // starts the moment its diagonal tile has its last update. With the leaves in the
    // common queue a leaf waited for a free workgroup -- all of them 9-us deep in the
    // trailing updates of the previous step -- 15 us out of every 72-us step. Both
    // lists are topological orders of the same graph, and a task only waits for tasks
    // that are earlier in the combined order: the earliest unfinished task of either
    // list is always claimed and runnable, whatever the residency of the grid.
    // (Members: the spine workgroups of all members come first in the grid, member by
    // member round robin, so that every member's first leaf starts at once.)
    const bool spine = (int)blockIdx.x < p.nspwg * nmem;
    const PTask *const list = spine ? p.spine : p.tasks;
    const int nlist = spine ? p.nspine : p.ntasks;
    const int spine_member = spine ? (int)blockIdx.x % nmem : 0;
    int spine_next = spine ? (int)blockIdx.x / nmem : 0;
    for (;;) {
        if (wave == 0) {
            int tc = spine_next;
            if (!spine && lane == 0)
                tc = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
            const int tg = __builtin_amdgcn_readfirstlane(tc);
            // position in the interleaved queue -> (member, task of the graph)
            const int member = spine ? spine_member : (nmem > 1 ? tg % nmem : 0);
            const int t = spine ? tg : (nmem > 1 ? tg / nmem : tg);
            int *cm = ctl + (long long)member * p.pstride;   // the member's counters
            int ab = 0;
            if (p.dbg && lane == 0) {
                p.dbg[8 * blockIdx.x + 0] = tg;
                p.dbg[8 * blockIdx.x + 1] = 1;
            }
            if (t < nlist) {
                const PTask *tk = list + t;
                const int ndep = __builtin_amdgcn_readfirstlane((int)tk->ndep);
                const long long t0 = wall_clock64();
                if (p.trace && lane == 0 && member == 0) {
                    const int ti = spine ? p.ntasks + t : t;
                    p.trace[32 * ti] = t0;
                    p.trace[32 * ti + 3] = blockIdx.x;
                }
                // All counters of the task are polled TOGETHER (round 5): lane i < ndep watches
                // dependency i, lane ndep the abort flag, one load instruction per round. Until
                // round 4 the dependencies were waited for one after the other, and a poll is a
                // round trip to the memory side (about 1.5 us) whether the counter has long
                // been there or not: three or four of them in front of EVERY task -- 4.5 us
                // before a 6-us K = 128 product, and 5.5 us on each of the two hops per tile of
                // the chain (products into the chain's tile, then the spine task).
                const bool is_dep = lane < ndep;
                const int *c = &ctl[2];
                int need = 1;                            // (abort flag: "reached" = set)
                if (is_dep) {
                    const int di = (int)tk->dep[lane];
                    const bool gate = di >= p.nctr;
                    c = gate ? p.gates + (di - p.nctr) : cm + PCTL_HEAD + di;
                    need = gate ? (di == p.nctr ? p.gate_need0 : p.gate_need1) : (int)tk->thr[lane];
                }
                if (p.dbg && lane == 0) {
                    p.dbg[8 * blockIdx.x + 2] = (int)(c - cm) - PCTL_HEAD;
                    p.dbg[8 * blockIdx.x + 3] = need;
                }
                for (;;) {
                    // (every lane loads: lanes beyond ndep watch the abort flag too)
                    const int v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long missing = __ballot(is_dep && v < need);
                    const unsigned long long stop = __ballot(!is_dep && v != 0);
                    if (missing == 0ull) break;
                    if (stop != 0ull || wall_clock64() - t0 > p.timeout) {
                        ab = 1;
                        break;
                    }
                }
            }
            if (lane == 0) {
                if (ab)
                    __hip_atomic_store(&ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_task = t;
                s_abort = ab;
                s_member = member;
            }
        }
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(s_task);
        if (t >= nlist || __builtin_amdgcn_readfirstlane(s_abort)) break;
        const int member = __builtin_amdgcn_readfirstlane(s_member);
        spine_next += p.nspwg;
        const int ti = spine ? p.ntasks + t : t;          // row of the debug trace
        // the tiles this task reads are complete at the memory side (see agent_load16);
        // nothing may be hoisted above the barrier
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (p.strict) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        const PTask &tk = list[t];
        const int op = __builtin_amdgcn_readfirstlane(tk.op);
        if (p.dbg && tid == 0) p.dbg[8 * blockIdx.x + 1] = 2;
        // (member-batched launches -- solo ones -- trace member 0)
        long long *tr = p.trace && member == 0 ? p.trace + 32 * ti : nullptr;
        if (tr && tid == 0) {
            tr[1] = wall_clock64();
            tr[12] = __builtin_amdgcn_s_memtime();
        }
        const long long mo = (long long)member * p.mstride;
        PanelCtx cx;
        cx.bA = p.bA + mo; cx.bW = p.bW + mo; cx.bX = p.bX + mo;
        cx.ctl = ctl + (long long)member * p.pstride; cx.gctl = ctl;
        cx.info = p.info + member; cx.timeout = p.timeout;
        cx.ld = p.ld; cx.goff = p.goff; cx.strict = p.strict; cx.leafskip = p.leafskip;
        if (op == PT_XS && !xs_run<false>(cx, &tk, tr)) break;
        if (op == PT_UF && !uf_run(cx, &tk, tr)) break;
        if (op == PT_LEAF || op == PT_UF || (op == PT_XS && tk.beta1 == 2)) {
            // F(s); behind an XS / UF task it is the leaf of the tile that task has just
            // updated, still in LDS (tile offCin, stream counter khi)
            const bool fused = op != PT_LEAF;
            const long long o = pt_off(fused ? tk.offCin : tk.offA, p.ld);
            const int cy = fused ? tk.khi : (tk.khi ? tk.klo : -1);
            run_leaf(cx, o, p.goff + tk.goff, cy, fused, tr);
        } else if (op != PT_XS) {
            run_gemm(cx, &tk);
        }
        // publish: every wave's (write-through) stores are acknowledged before the
        // counter moves
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (p.strict && tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (p.dbg && tid == 0) p.dbg[8 * blockIdx.x + 1] = 3;
        if (tr && tid == 0) {
            tr[2] = wall_clock64();
            tr[13] = __builtin_amdgcn_s_memtime();
        }
        if (tid == 0)
            __hip_atomic_fetch_add(cx.ctl + PCTL_HEAD + tk.sig, (int)tk.siginc, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }

    if (p.dbg && tid == 0) p.dbg[8 * blockIdx.x + 1] = 4;
    if (tid == 0) {
        const int gone = __hip_atomic_fetch_add(&ctl[1], 1, __ATOMIC_ACQ_REL,
                                                __HIP_MEMORY_SCOPE_AGENT);
        s_task = gone == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (s_task) {                               // last one out: leave clean blocks
        if (__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            for (int m = tid; m < nmem; m += 256) atomicCAS(p.info + m, 0, -1);
        __syncthreads();
        // (every thread: 3 100 counters for a whole 4096-matrix)
        for (int m = 0; m < nmem; ++m)
            for (int i = tid; i < PCTL_HEAD + p.nctr; i += 256)
                __hip_atomic_store(&ctl[(long long)m * p.pstride + i], 0, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- lock-step sweep over many members (round 4) ---------------------------------
// With dozens of members in a group the parallelism comes from the members, not from
// overlapping the steps of one factorisation: the same tile tasks -- the leaf, the
// row-panel solves XS(s,t), the fused XSF(s+1) = solve + diagonal update + leaf -- run as
// plain launches, one phase of the factorisation at a time over ALL members (workgroup =
// one task of one member, kernel boundaries instead of counters), and the trailing updates
// between the phases go to the tile engine, batched over the members, where they run at
// MFMA rate instead of as 64 x 64 tasks of lone workgroups bound by what one CU can request
// through agent-scope accesses (the member-batched panel launch spends 2.9 ms on 32
// matrices of order 1024, three times its tasks' time alone). Same task bodies, same
// order of accumulation per element: the bits of the panel launch.
struct SweepArgs {
    double *bA, *bW, *bX;                    // member 0's block origins
    int ld;
    long long mstride;
    const PTask *tasks;                      // this phase's tasks of ONE member (XSF first)
    int ntasks, nmem;
    int *ctl;                                // a constant control block: "everything is published"
    int *info;
    int goff;
    long long timeout;
    int strict, leafskip;
    int presolved;                           // the fused tasks' tiles come solved (sweep_xs_kernel)
};

__global__ __launch_bounds__(256) void sweep_kernel(SweepArgs p)
{
    // the fused tasks of all members first: they are the longest of a phase
    const int member = (int)blockIdx.x % p.nmem;
    const PTask &tk = p.tasks[(int)blockIdx.x / p.nmem];
    const int op = __builtin_amdgcn_readfirstlane(tk.op);
    const long long mo = (long long)member * p.mstride;
    PanelCtx cx;
    cx.bA = p.bA + mo; cx.bW = p.bW + mo; cx.bX = p.bX + mo;
    cx.ctl = p.ctl; cx.gctl = p.ctl;
    cx.info = p.info + member; cx.timeout = p.timeout;
    cx.ld = p.ld; cx.goff = p.goff; cx.strict = p.strict; cx.leafskip = p.leafskip;
    if (op == PT_XS) {
        const bool pre = __builtin_amdgcn_readfirstlane(p.presolved) != 0;
        if (pre ? !xs_run<true>(cx, &tk, nullptr) : !xs_run<false>(cx, &tk, nullptr)) return;
    }
    if (op == PT_LEAF || (op == PT_XS && tk.beta1 == 2)) {
        const bool fused = op == PT_XS;
        run_leaf(cx, pt_off(fused ? tk.offCin : tk.offA, p.ld), p.goff + tk.goff, -1, fused,
                 nullptr);
    }
}

// ---- lock-step sweep: the row-panel tiles as a DENSE launch (round 5, second half) -----
// What binds a group of many members at small orders is occupancy, not arithmetic
// (profiles/r05_small_groups.txt): every XS task of sweep_kernel holds a whole CU -- the
// launch asks for the leaf's 157 KB of LDS -- for 26 us while it uses the matrix pipe for 8,
// and the trailing updates between the phases are rank-128 .. 896 products that move more
// bytes than they compute on (N = 512: 1 536 workgroups for 48 us at K = 128, 6 TB/s of
// operand and C traffic; N = 2048: the tile engine reaches 30 TFLOP/s there beside the other
// group's sweep phases). This kernel runs the solves of tile row s for the tiles t >= t0 with
// the tile in REGISTERS from the first load on (the accumulator layout of xs_run: two
// 16-column strips a wave) and 77 KB of LDS -- two workgroups a CU, each other's latencies
// covered -- and it applies the trailing updates of its tile ITSELF first: the steps
// kfirst .. s-1 (all of them by default), X -= R(k,s)^T R(k,t), rank 16 at a time through a
// double-buffered staging area, exactly the fold of xs_run (k ascending in the groups of four
// of the tile engine's MFMA steps, -(a) b accumulated into X: the same bits as the product it
// replaces). Update-only workgroups bring the next diagonal tile (and, right-looking, every
// tile of the rows below) up to date in the same launch; tiles of the right-hand-side column
// carry one live strip. What is left for sweep_kernel is the last diagonal update and the
// leaf (xs_run<PRE>): two launches a tile row, no product of the tile engine between them.
#define XL_LS 144                            // staging row stride (doubles): 16 mod 32 -- the two
                                             // row groups of a half-wave hit disjoint banks
#define XSL_STAGE (4 * 16 * XL_LS)           // doubles: two stages x two operands x 16 rows
// ... overlaid, for the solve, by ALL of R_ss right of its diagonal 16-blocks (panel pp: 16 rows
// of 112 - 16 pp doubles, row stride 114 - 16 pp) and the eight inverses of those blocks
#define XSL_RP(pp) (16 * (114 * (pp) - 8 * (pp) * ((pp) - 1)))
#define XSL_YP(pp) (XSL_RP(8) + 16 * YS * (pp))
#define XSL_LDS ((XSL_YP(8) > XSL_STAGE ? XSL_YP(8) : XSL_STAGE) * 8)   // 76 800 B: two a CU
struct SweepXsArgs {
    double *bA, *bW, *bX;                    // member 0's block origins
    int ld;
    long long mstride;
    int nmem;
    int s, t0, kfirst;                       // tile row, first tile column, first folded step
    int kfirst_narrow;                       // first folded step of the tiles of narrow_col
    int narrow_col;                          // tile column whose tiles carry ONE meaningful column
                                             // (the right-hand side, column 0): only the first
                                             // 16-column strip of such a tile is computed; -1: none
    int TW, right;                           // tile columns (with the right-hand side); right = 1:
                                             // the update-only workgroups are ALL tiles below
                                             // row s (right-looking: step s-1 everywhere)
    int nsolve;                              // tiles t0 .. t0 + nsolve - 1 are updated and solved;
                                             // two more workgroups per member (if the grid has
                                             // them) only UPDATE: tile (s, s+1) and the diagonal
                                             // tile (s+1, s+1), which the fused task then takes
    long long *trace;                        // GPX_XS_DEBUG: stamps of workgroup 0 (else null)
};

__global__ __launch_bounds__(256, 2) void sweep_xs_kernel(SweepXsArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int member = (int)blockIdx.x % p.nmem;
    const int idx = (int)blockIdx.x / p.nmem;
    // role 0: update + solve tile (s, t), staging matrix -> R. Update-only workgroups (role 1):
    // tile (urow, t) in place -- in A if it is a diagonal tile, else in the staging matrix --
    // with R(k, urow)^T R(k, t): tile (s, s+1) and / or the next diagonal tile, or (right) every
    // tile of the rows below s
    const int nupd = (int)gridDim.x / p.nmem - p.nsolve;
    const int role = idx < p.nsolve ? 0 : 1;
    const int s = p.s, ld = p.ld;
    int urow = s, t = p.t0 + idx;
    if (role) {
        int j = idx - p.nsolve;
        if (p.right) {
            urow = s + 1;
            for (int len = p.TW - urow; j >= len; len = p.TW - urow) {
                j -= len;
                ++urow;
            }
            t = urow + j;
        } else if (nupd == 2 && j == 0) {
            t = s + 1;
        } else {
            urow = s + 1;
            t = s + 1;
        }
    }
    const bool in_a = role && t == urow;                 // a diagonal tile lives in A
    const long long mo = (long long)member * p.mstride;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    // A tile of the right-hand-side column: a = R^-T (y - m) is its column 0, the other 127 are
    // zero on the way in and read by nobody on the way out. Its updates and its solve cost what
    // a full tile's cost -- a quarter of all update work at N = 1024, a third at N = 512 -- so
    // only the strip that holds column 0 is loaded, computed and stored (wave 0, its first
    // strip: an eighth of the MFMAs; the chains of column 0 are untouched: the same bits).
    const bool narrow = t == p.narrow_col;
    const bool idle = narrow && wave != 0;               // (wave-uniform) nothing of X to do
    const bool two = !narrow;                            // the wave's second strip is live
    double *St = reinterpret_cast<double *>(smem_raw);   // [2][2][16][XL_LS]
    const double *Am = p.bA + mo;

    // (all global accesses as buffer instructions: one 32-bit lane offset each, everything
    // that is wave-uniform -- the row block, the step -- in the scalar offset; with 64-bit
    // addresses per access the update loop spilled its prefetch registers)
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    auto load8 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    };
    auto load16 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) -> double2 {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    // tile in, straight into the accumulator layout: xr[cc][q][r] = X[16q + lk + 4r][c0 + lr]
    v4d xr[2][NBK];
    const int vtile = (lk * ld + 16 * wave + lr) * 8;    // lane part of an element of the layout
    {
        __amdgpu_buffer_rsrc_t rX = agent_rsrc((in_a ? p.bA : p.bX) + mo +
                                               (long long)(LB * urow) * ld + (long long)LB * t);
        if (!idle) {
#pragma unroll
            for (int q = 0; q < NBK; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int soff = (16 * q + 4 * r) * ld * 8;
                    xr[0][q][r] = load8(rX, vtile, soff);
                    if (two) xr[1][q][r] = load8(rX, vtile + 512, soff);
                }
        }
    }

    long long *tr = p.trace && blockIdx.x == 0 && tid == 0 ? p.trace : nullptr;
    if (tr) {
        tr[0] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[1] = wall_clock64();                          // tile in
    }
    // trailing updates kfirst .. s-1 of this tile: 16 rows of R(k,s) and R(k,t) a step; the
    // row blocks of steps kfirst .. s-1 are consecutive rows of tile columns s and t
    const int kfirst = narrow ? p.kfirst_narrow : p.kfirst;
    const int nst = 8 * (s - kfirst);
    if (nst > 0) {
        __amdgpu_buffer_rsrc_t rFA = agent_rsrc(Am + (long long)(LB * kfirst) * ld +
                                                (long long)LB * urow),
                               rFB = agent_rsrc(Am + (long long)(LB * kfirst) * ld + (long long)LB * t);
        const int vrow = (wave * ld + 2 * lane) * 8, rstep = 4 * ld * 8, sstep = 16 * ld * 8;
        double2 fa[2][4], fb[2][4];
        auto issue_f = [&](int set, int st) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[set][i] = load16(rFA, vrow + i * rstep, st * sstep);
                fb[set][i] = load16(rFB, vrow + i * rstep, st * sstep);
            }
        };
        auto stage_f = [&](int par) {
            double *stA = St + 32 * par * XL_LS, *stB = stA + 16 * XL_LS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<double2 *>(stA + (wave + 4 * i) * XL_LS + 2 * lane) = fa[par][i];
                *reinterpret_cast<double2 *>(stB + (wave + 4 * i) * XL_LS + 2 * lane) = fb[par][i];
            }
        };
        auto mfma_f = [&](int par) {
            if (idle) return;
            const double *stA = St + 32 * par * XL_LS, *stB = stA + 16 * XL_LS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double *ra = stA + (4 * ks + lk) * XL_LS + lr;
                const double *rb = stB + (4 * ks + lk) * XL_LS + 16 * wave + lr;
                double a[NBK];
#pragma unroll
                for (int q = 0; q < NBK; ++q) a[q] = -ra[16 * q];
                const double b0 = rb[0], b1 = rb[64];
#pragma unroll
                for (int q = 0; q < NBK; ++q)
                    xr[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b0, xr[0][q], 0, 0, 0);
                if (two) {                               // (one wave-uniform branch a k-step)
#pragma unroll
                    for (int q = 0; q < NBK; ++q)
                        xr[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1, xr[1][q], 0, 0, 0);
                }
            }
        };
        // the rows of steps st + 1 and st + 2 are in flight while the products of step st run
        // (a step is 1.8 us of MFMA, a row block takes longer than that to arrive); one barrier
        // a step: register set and LDS stage alternate together
        issue_f(0, 0);
        issue_f(1, 1);
#pragma unroll 1
        for (int st = 0; st < nst; st += 2) {            // (nst is a multiple of 8)
            stage_f(0);
            if (st + 2 < nst) issue_f(0, st + 2);
            __syncthreads();
            mfma_f(0);
            stage_f(1);
            if (st + 3 < nst) issue_f(1, st + 3);
            __syncthreads();
            mfma_f(1);
        }
        __syncthreads();                                 // the staging area becomes the panel buffers
    }

    // the solve against R_ss, panel by panel (xs_run's arithmetic; R_ss and the inverses of its
    // diagonal 16-blocks are final: read from R and W themselves, two panels ahead)
    __amdgpu_buffer_rsrc_t rR = agent_rsrc(Am + (long long)(LB * s) * ld + (long long)LB * s),
                           rW = agent_rsrc(p.bW + mo + (long long)(LB * s) * ld + (long long)LB * s);
    __amdgpu_buffer_rsrc_t rO = agent_rsrc((role && !in_a ? p.bX : p.bA) + mo +
                                           (long long)(LB * urow) * ld + (long long)LB * t);
    auto rows_out = [&](int pb) {                        // row block pb of R_st, from the registers
        if (idle) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int soff = (16 * pb + 4 * r) * ld * 8;
            // (through locals: __builtin_bit_cast of a vector ELEMENT took element 0 every time)
            const double v0 = xr[0][pb][r];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, v0), rO, vtile, soff, 0);
            if (two) {
                const double v1 = xr[1][pb][r];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, v1), rO, vtile + 512,
                                                      soff, 0);
            }
        }
    };
    if (tr) tr[2] = wall_clock64();                      // updates done
    if (role != 0) {                                     // (workgroup-uniform) updated: back it goes
#pragma unroll
        for (int pb = 0; pb < NBK; ++pb) rows_out(pb);
        return;
    }
    // All eight panels go to LDS BEFORE the first step (two rounds of loads, one barrier): a step
    // is then LDS reads and MFMAs only -- 288 MFMAs a wave in all, 4 us -- where the
    // panel-by-panel form of xs_run (which has to follow a running leaf) spends two barriers
    // and an LDS turn-around per step, 18-25 us a tile. Same operands, same order: same bits.
    // (the second half of the panels is in flight while the first four steps run)
    XsPanelRegs g[4];
    auto commit_h = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pp = 4 * h + i, ncol2 = 56 - 8 * pp, slots = 16 * ncol2, xs = 114 - 16 * pp;
            double *Rq = St + XSL_RP(pp), *Yq = St + XSL_YP(pp);
            if (tid < 128) {
                const int r = tid >> 3, c = tid & 7;         // W[r][2c..]: Y[k][r]
                Yq[(2 * c) * YS + r] = g[i].y.x;
                Yq[(2 * c + 1) * YS + r] = g[i].y.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < xs_nload(pp)) {
                    const int e = tid + 256 * j, r = e / ncol2, c2 = e % ncol2;
                    if (e < slots) *reinterpret_cast<double2 *>(Rq + r * xs + 2 * c2) = g[i].r[j];
                }
        }
    };
    auto step = [&](int pp) {
        const double *Rp = St + XSL_RP(pp), *Yp = St + XSL_YP(pp);
        const int xs = 114 - 16 * pp;
        if (tr) tr[8 + pp] = wall_clock64();
        if (idle) return;
        if (pp >= 1) rows_out(pp - 1);
        // X[p] <- Y_p X[p]
        v4d xn[2];
        {
            double ya[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ya[r] = Yp[lr * YS + lk + 4 * r];   // Y[i][k]
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                if (cc == 1 && !two) continue;
                v4d tt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tt = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[r], xr[cc][pp][r], tt, 0, 0, 0);
                xr[cc][pp] = tt;
                xn[cc] = tt;
            }
        }
        // X[q] -= R[p][q]^T X[p], q > p: the first strips of all q, then the second ones
        double ra[NBK][4];
#pragma unroll
        for (int q = pp + 1; q < NBK; ++q) {
            const double *rq = Rp + 16 * (q - pp - 1) + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) ra[q][r] = -rq[(lk + 4 * r) * xs];   // -R[p][q][k][i]
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xr[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[q][r], xn[0][r], xr[0][q], 0, 0, 0);
        }
        if (two) {
#pragma unroll
            for (int q = pp + 1; q < NBK; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xr[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[q][r], xn[1][r], xr[1][q], 0, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) xs_issue(g[i], rR, rW, ld, i, tid);
    commit_h(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) xs_issue(g[i], rR, rW, ld, 4 + i, tid);
    __syncthreads();
    if (tr) tr[3] = wall_clock64();                      // panels 0-3 staged
    step(0); step(1); step(2); step(3);
    commit_h(1);
    __syncthreads();
    step(4); step(5); step(6); step(7);
    rows_out(NBK - 1);
    if (tr) {
        tr[4] = wall_clock64();                          // last step issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[5] = wall_clock64();                          // stores acknowledged
    }
}

// ---- host: task graph of one panel ---------------------------------------------
namespace {

struct Graph {
    int T, ld;
    int E = 0;                                     // wide panel: E more tile columns (and the
                                                   // E x E diagonal block below them) right of
                                                   // the block take its row panel and update
    bool stream;                                   // XS tasks beside the leaves (default)
    int kbatch = 1;                                // steps per trailing-update task of a far tile
    bool aug = false;                              // E = 1 tile column right of a WHOLE matrix:
                                                   // a right-hand side (no block below it, no gates)
    bool fold = false;                             // (with split) the spine's solves fold the last
                                                   // update of their tile in themselves
    bool split = false;                            // round 5: the diagonal update and the leaf of
                                                   // tile t on a workgroup of their own (PT_UF)
                                                   // that follows the spine's solve of (t-1, t)
    int ig = PANEL_IG;                             // tiles per inverse group (T: the launch leaves
                                                   // the whole of R^-1 behind)
    // cost of a solve / follower of the chain in the schedule's simulation: they overlap (each
    // follows the rows of the one before) but the simulation runs them end to end, so their
    // nominal 38 / 60 us make the simulated chain slower than the real one and the queue holds
    // the chain's feeder tasks back behind bulk work; much shorter and the workers sit in
    // claimed tasks that are not ready. Scale 0.5 / 0.6 / 0.7 / 0.8 / 1.0 / 1.4 / 2.0: N = 4096
    // value-only 1.34 / 1.29 / 1.29 / 1.295 / 1.32 / 1.39 / 1.39 ms, with gradients 2.22 / 2.155
    // / 2.15 / 2.155 / 2.165 / 2.27 / 2.26; N <= 3072 within noise.
    static double chain_us(double us)
    {
        static const double f = getenv("GPX_PANEL_CHAIN_SCALE") ? atof(getenv("GPX_PANEL_CHAIN_SCALE")) : 0.75;
        return us * f;
    }
    int inv_chunks(int i, int s) const             // stages of the scratch tile (i, s)
    {
        return ig > PANEL_IG ? (s - 1 - i + 3) / 4 + 1 : 1;   // bulk chunks of four tile rows + the last row
    }
    std::vector<PTask> tasks;
    std::vector<double> cost;                      // microseconds, for the schedule
    std::vector<double> early;                     // when sig2 fires after the start (or < 0)
    std::vector<std::vector<int>> signalers;       // per counter, generation order
    std::vector<std::vector<int>> sigcum;          // count after that task's signal

    int TW() const { return T + E; }
    int cA(int s, int t) const { return s * TW() + t; }
    int cX(int i, int j) const { return TW() * TW() + i * T + j; }
    int cW(int i, int j) const { return TW() * TW() + T * T + i * T + j; }
    int cY(int s) const { return TW() * TW() + 2 * T * T + s; }  // row panels of R_ss published by F(s)
    int nctr() const { return TW() * TW() + 2 * T * T + T; }
    long long tile(int s, int t) const { return (long long)(128 * s) * ld + 128 * t; }
    long long sub(int s, int t, int a, int b) const
    {
        return tile(s, t) + (long long)(SUB * a) * ld + SUB * b;
    }
    // Counter units: a 64x64 product task adds U = 4, a 32x32 one adds 1, a leaf adds
    // a whole stage of its tile (4 U); a tile has finished a stage (its row panel, or one
    // trailing update) after 4 U.
    static constexpr int U = 4, STAGE = 4 * U;
    static int r_ready(int s) { return STAGE * (s + 1); }   // counter value: R_st final

    PTask blank() const
    {
        PTask t;
        memset(&t, 0, sizeof(t));
        t.sub = SUB;
        t.sig2 = -1;
        return t;
    }
    void dep(PTask &t, int ctr, int thr)
    {
        if (thr <= 0) return;
        t.dep[t.ndep] = (short)ctr;
        t.thr[t.ndep] = (short)thr;
        t.ndep++;
    }
    void set_fold(PTask &k, int s, int t)              // the solve of tile (s, t) folds update s-1 in
    {
        k.fold = 1;
        k.offFA = tile(s - 1, s);
        k.offFB = tile(s - 1, t);
        // (host only: the two solves whose rows it polls)
        k.dep[k.ndep] = (short)cA(s - 1, s);
        k.thr[k.ndep] = (short)r_ready(s - 1);
        k.dep[k.ndep + 1] = (short)cA(s - 1, t);
        k.thr[k.ndep + 1] = (short)r_ready(s - 1);
        k.nhost = 2;
    }
    void push(PTask t, int ctr, int inc, double us)
    {
        t.sig = (short)ctr;
        t.siginc = (short)inc;
        const int id = (int)tasks.size();
        tasks.push_back(t);
        cost.push_back(us);
        early.push_back(-1.0);
        const int before = sigcum[ctr].empty() ? 0 : sigcum[ctr].back();
        signalers[ctr].push_back(id);
        sigcum[ctr].push_back(before + inc);
    }
    // measured task times (GPX_PANEL_DEBUG=2): 64-tiles 5.4 us at K = 64, 9.2 at 128,
    // 46 at 896; 32-tiles about half of that
    static double gemm_us(int klo, int khi, int sub = SUB)
    {
        return sub == SUB ? 3.0 + 0.75 * ((khi - klo) / 16) : 2.5 + 0.35 * ((khi - klo) / 16);
    }

    void build()
    {
        signalers.assign(nctr(), {});
        sigcum.assign(nctr(), {});
        // Wide panel (round 3): the tile columns T .. T+E-1 right of the block get their
        // row-panel tiles R_st by the same XS tasks (beside the leaf of their row) and the
        // trailing updates reach them and the E x E diagonal block below them: when the
        // launch ends, R[k, k+1] and update k of block (k+1, k+1) are done and the next
        // panel starts at once. Before, these were two dependent launches on the chain of
        // diagonal blocks (a triangle-aware product with W_kk^T and a SYRK, 60-140 us
        // each between two 360-us panels at N = 4096).
        const int TWc = TW();
        for (int s = 0; s < T; ++s) {
            if (s == 0 || !stream) {   // F(s); when streaming, F(s > 0) is the tail of XSF(s)
                PTask k = blank();
                k.op = PT_LEAF;
                k.offA = tile(s, s);
                k.offB = tile(s, s);
                k.goff = 128 * s;
                if (stream) {
                    k.klo = cY(s);
                    k.khi = 1;
                }
                dep(k, cA(s, s), STAGE * s);
                push(k, cA(s, s), STAGE, 40.0);
            }
            // inverse column s (needs only R_{s-1,s} and the previous columns), inside the
            // 1024-block of tile s: a launch over a whole matrix (round 3, T up to 32)
            // leaves behind what the blocked driver leaves, R and the inverses of its
            // diagonal blocks
            for (int i = s / ig * ig; i < s; ++i) {
                if (ig > PANEL_IG) {
                    // The WHOLE inverse in this launch (round 5, with gradients in view). The
                    // sum over k is cut so that what column s has to WAIT for is short: the
                    // tile rows i .. s-2 in bulk chunks of four tiles (K <= 512) that pass the
                    // partial sum on through the scratch tile and can run a step early -- they
                    // need W(i, s-2) and R(s-2, s) --, then the one term that needs the column
                    // before, W(i, s-1) R(s-1, s), as four 64x64 tasks with K = 128. The
                    // columns of the inverse are a chain of their own (column s needs
                    // W(i, s-1)): with the last FOUR tile rows in its link the chain ran at
                    // 40 us a column with 64-tiles and 60-100 us with 128-tiles, behind the
                    // chain of the factorisation's 31. Bulk chunks but the last are throughput
                    // work: one 128x128 task each (GPX_PANEL_I128=0: four 64x64), half the
                    // operand bytes per flop, and a lone workgroup is bound by the bytes it
                    // can request.
                    const int nb = inv_chunks(i, s) - 1;              // bulk chunks
                    // (from 17 tiles on: up to 16 the workers have time to spare and the
                    // shorter tasks win, N = 2048 0.83 against 0.87 ms)
                    static const int big_env = getenv("GPX_PANEL_I128") ? atoi(getenv("GPX_PANEL_I128")) : -1;
                    const bool big = big_env >= 0 ? big_env != 0 : T > 16;
                    for (int c = 0; c < nb; ++c) {
                        const int l = std::min(i + 4 * c + 3, s - 2);     // last tile row of the chunk
                        // (the last bulk chunk needs W(i, s-2), one step old: four short tasks)
                        const bool one = big && c < nb - 1;
                        const int ntask = one ? 1 : 4;
                        for (int q = 0; q < ntask; ++q) {
                            const int a = q >> 1, bq = q & 1, e = one ? 128 : SUB;
                            PTask k = blank();
                            k.op = PT_GEMM_NN;
                            k.sub = (short)e;
                            k.bufA = 1; k.offA = tile(i, i) + (long long)(e * a) * ld;
                            k.bufB = 0; k.offB = tile(i, s) + e * bq;
                            const long long oc = tile(i, s) + (long long)(e * a) * ld + e * bq;
                            k.bufCin = 2; k.offCin = oc;
                            k.bufCout = 2; k.offCout = oc;
                            // (W_ii upper: k >= row start; a 128-tile's lower rows start at
                            // k = 0 too, W_ii holds zeros there)
                            k.klo = c == 0 ? e * a / 64 * 64 : 512 * c;
                            k.khi = 128 * (l + 1 - i);
                            k.beta1 = c > 0;
                            if (l == i) dep(k, cA(i, i), STAGE * (i + 1));
                            else dep(k, cW(i, l), STAGE);
                            dep(k, cA(l, s), r_ready(l));
                            dep(k, cX(i, s), STAGE * c);
                            if (one) push(k, cX(i, s), STAGE, 6.0 + 2.2 * ((k.khi - k.klo) / 16));
                            else push(k, cX(i, s), U, gemm_us(k.klo, k.khi));
                        }
                    }
                    for (int a = 0; a < 2; ++a)
                        for (int bq = 0; bq < 2; ++bq) {     // the last term: tile row s-1
                            PTask k = blank();
                            k.op = PT_GEMM_NN;
                            k.bufA = 1; k.offA = tile(i, i) + (long long)(SUB * a) * ld;
                            k.bufB = 0; k.offB = tile(i, s) + SUB * bq;
                            const long long oc = tile(i, s) + (long long)(SUB * a) * ld + SUB * bq;
                            k.bufCin = 2; k.offCin = oc;
                            k.bufCout = 2; k.offCout = oc;
                            k.klo = 128 * (s - 1 - i) + (s - 1 == i ? SUB * a : 0);
                            k.khi = 128 * (s - i);
                            k.beta1 = nb > 0;
                            if (s - 1 == i) dep(k, cA(i, i), STAGE * (i + 1));
                            else dep(k, cW(i, s - 1), STAGE);
                            dep(k, cA(s - 1, s), r_ready(s - 1));
                            dep(k, cX(i, s), STAGE * nb);
                            push(k, cX(i, s), U, gemm_us(k.klo, k.khi));
                        }
                } else {
                // I1(i,s): T = W[i,i..s-1] R[i..s-1,s]. A lone workgroup loads ~23 GB/s, a
                // 64x64 tile with K = 896 takes 44 us and the long ones end up as the tail
                // of the launch: from K = 512 on in 32x32 tiles (half the bytes each)
                const int fine = (stream && 128 * (s - i) >= 512) ? 32 : SUB, nsub = 128 / fine;
                for (int a = 0; a < nsub; ++a)
                    for (int b = 0; b < nsub; ++b) {
                        PTask k = blank();
                        k.op = PT_GEMM_NN;
                        k.sub = (short)fine;
                        k.bufA = 1; k.offA = tile(i, i) + (long long)(fine * a) * ld;
                        k.bufB = 0; k.offB = tile(i, s) + fine * b;
                        const long long oc = tile(i, s) + (long long)(fine * a) * ld + fine * b;
                        k.bufCin = 2; k.offCin = oc;
                        k.bufCout = 2; k.offCout = oc;
                        k.klo = fine * a / 64 * 64;        // W_ii upper: k >= row start
                        k.khi = 128 * (s - i);
                        if (s - 1 == i) dep(k, cA(i, i), STAGE * (i + 1));
                        else dep(k, cW(i, s - 1), STAGE);
                        dep(k, cA(s - 1, s), r_ready(s - 1));
                        push(k, cX(i, s), fine == SUB ? U : 1, gemm_us(k.klo, k.khi, fine));
                    }
                }
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {       // I2(i,s): W_is = -T W_ss
                        PTask k = blank();
                        k.op = PT_GEMM_NN;
                        k.bufA = 2; k.offA = tile(i, s) + (long long)(SUB * a) * ld;
                        k.bufB = 1; k.offB = tile(s, s) + SUB * b;
                        k.bufCin = 1; k.offCin = sub(i, s, a, b);
                        k.bufCout = 1; k.offCout = sub(i, s, a, b);
                        k.klo = 0;
                        k.khi = SUB * (b + 1);
                        k.neg = 1;
                        dep(k, cX(i, s), STAGE * inv_chunks(i, s));
                        dep(k, cA(s, s), STAGE * (s + 1));
                        push(k, cW(i, s), U, gemm_us(k.klo, k.khi));
                    }
            }
            // row panel P(s,t). The tile right of the diagonal feeds the next leaf: it
            // is cut into 16 tasks of 32x32 (half the operand bytes per workgroup -- a
            // lone workgroup is bound by what one CU can request -- and a quarter of the
            // MFMA work), the others into 4 of 64x64
            // Streaming: one XS task per tile instead, working beside F(s) (it waits for
            // what F(s) waits for and then follows cY(s)); the one right of the diagonal
            // also applies the update of the next diagonal tile.
            // The tile right of the diagonal is XSF(s+1), a spine task: solve, update the
            // next diagonal tile in LDS and factor it there (F(s+1)), publishing cY(s+1).
            // R_{s-1,s} out (XSF(s)'s early signal) stands for "F(s) is about to start".
            for (int t = s + 1; stream && t < TWc; ++t) {
                PTask k = blank();
                k.op = PT_XS;
                k.bufA = 2;                              // R_ss's panels: from the tile's mailbox
                k.offA = tile(s, s);
                k.bufB = 2; k.offB = tile(s, t);
                k.bufCout = 0; k.offCout = tile(s, t);
                k.klo = cY(s);
                // fold: a solve of tile row s >= 1 starts with its tile at update s-2 and applies
                // update s-1 itself, following the rows of R(s-1,s) and R(s-1,t) (xs_run); the
                // products of that update leave the graph. The first two tiles of a row run on
                // the spine's workgroups (their rows are what the next row's spine follows).
                const bool sp2 = fold && split && t == s + 2 && t < T;
                const bool fol = fold && split && s >= 1;   // (every solve of the rows below the first)
                if (s > 0 && !fol) dep(k, cA(s - 1, s), STAGE * s);
                dep(k, cA(s, t), STAGE * (fol ? s - 1 : s));
                // an extra tile must carry the updates of the blocks before this one, which
                // another stream may still be applying when the launch starts (gate 0: the
                // tiles of this block's rows); later rows inherit the order through cA
                if (t >= T && s == 0 && !aug) dep(k, nctr() + 0, 1);
                // Split spine: a solve of tile row s follows the leaf of tile s, which runs
                // inside UF(s) -- and UF(s) only starts when its diagonal tile has every update
                // before step s-1. "R_{s-1,s} is out" no longer says that (the solving task
                // does not touch the diagonal tile), so the solves wait for it themselves:
                // without this the queue may hold all of row s in front of the product that
                // UF(s) waits for, every worker claims a solve of row s and spins for a leaf
                // that cannot start (seen at T = 32: 250 workers on row 10, UF(10) waiting
                // for update 8 of its tile).
                // (with the fold UF(s) applies step s-2 itself and starts at update s-3)
                if (split && s >= 1) dep(k, cA(s, s), STAGE * (fold ? s - 2 : s - 1));
                if (t == s + 1 && t < T && split) {
                    // the spine solves the tile (a plain row-panel task of the spine's list,
                    // between the follower tasks of tiles s and s+1) ...
                    k.spine = 1;
                    k.goff = 128 * t - 64;               // (sort key only)
                    if (fol) set_fold(k, s, t);
                    // (a folding task stands for the update it applied as well)
                    push(k, cA(s, t), (fol ? 2 : 1) * STAGE, chain_us(38.0));
                    // ... and UF(t) on another spine workgroup follows its rows
                    PTask u = blank();
                    u.op = PT_UF;
                    u.bufA = 0; u.offA = tile(s, t);
                    u.bufCin = 0; u.offCin = tile(t, t);
                    u.khi = cY(t);
                    u.goff = 128 * t;
                    // (with the fold it applies update s-1 of its tile as well, following the
                    // rows of R(s-1,t): uf_run)
                    const bool uf2 = fold && s >= 1;
                    dep(u, cA(t, t), STAGE * (uf2 ? s - 1 : s));
                    // (host only, for the order and the checks: it polls the rows of R_st)
                    u.dep[u.ndep] = (short)cA(s, t);
                    u.thr[u.ndep] = (short)r_ready(s);
                    u.nhost = 1;
                    if (uf2) {
                        u.fold = 1;
                        u.offFA = tile(s - 1, t);
                        u.dep[u.ndep + 1] = (short)cA(s - 1, t);
                        u.thr[u.ndep + 1] = (short)r_ready(s - 1);
                        u.nhost = 2;
                    }
                    // updates s-1 (folded) and s, then R_tt and W_tt
                    push(u, cA(t, t), (uf2 ? 3 : 2) * STAGE, chain_us(60.0));
                } else if (sp2) {
                    // the second tile of the row on the spine too (between the first and UF(s+1))
                    k.spine = 1;
                    k.goff = 128 * (s + 1) - 32;
                    if (fol) set_fold(k, s, t);
                    push(k, cA(s, t), (fol ? 2 : 1) * STAGE, chain_us(38.0));
                } else if (t == s + 1 && t < T) {
                    k.beta1 = 2;
                    k.bufCin = 0; k.offCin = tile(t, t);
                    k.khi = cY(t);
                    k.goff = 128 * t;
                    k.sig2 = (short)cA(s, t);
                    dep(k, cA(t, t), STAGE * s);
                    const int id = (int)tasks.size();
                    push(k, cA(t, t), 2 * STAGE, 85.0);   // update s, then R_tt and W_tt
                    early[id] = 42.0;
                    const int before = sigcum[cA(s, t)].empty() ? 0 : sigcum[cA(s, t)].back();
                    signalers[cA(s, t)].push_back(id);
                    sigcum[cA(s, t)].push_back(before + STAGE);
                } else {
                    if (fol) set_fold(k, s, t);
                    push(k, cA(s, t), (fol ? 2 : 1) * STAGE, chain_us(38.0));
                }
            }
            for (int t = s + 1; !stream && t < T; ++t) {
                const int fine = (t == s + 1) ? 32 : SUB, nsub = 128 / fine;
                for (int a = 0; a < nsub; ++a)
                    for (int b = 0; b < nsub; ++b) {
                        PTask k = blank();
                        k.op = PT_GEMM_TN;
                        k.sub = (short)fine;
                        k.bufA = 1; k.offA = tile(s, s) + fine * a;
                        k.bufB = 2; k.offB = tile(s, t) + fine * b;
                        const long long oc = tile(s, t) + (long long)(fine * a) * ld + fine * b;
                        k.bufCin = 0; k.offCin = oc;
                        k.bufCout = 0; k.offCout = oc;
                        k.klo = 0;
                        k.khi = (fine * (a + 1) + 63) / 64 * 64;   // W_ss upper: k < column end
                        dep(k, cA(s, s), STAGE * (s + 1));
                        dep(k, cA(s, t), STAGE * s);
                        push(k, cA(s, t), fine == SUB ? U : 1, gemm_us(k.klo, k.khi, fine));
                    }
            }
            // trailing update S(s,q,t), next diagonal tile first (and in 32x32 tasks)
            for (int q = s + 1; q < TWc; ++q)
                for (int t = q; t < TWc; ++t) {
                    if (stream && q == s + 1 && t == s + 1 && q < T) continue;   // inside XSF(s+1)
                    if (fold && split && q == s + 1 && t > q)
                        continue;                       // folded in by the solves of row s+1
                    if (fold && split && q == s + 2 && t == q && q < T)
                        continue;                       // folded in by UF(s+2)
                    if (aug && q >= T) continue;            // nothing below a right-hand side
                    // A whole matrix (T > 8): the updates a tile takes long before its own
                    // row is due -- steps up to q - 3 -- are batched, `kbatch` steps per task
                    // (one read and one write of the tile for kbatch x 128 of k: the launch is
                    // bound by what its tasks move through agent-scope accesses). The two
                    // updates right before the tile's row stay tasks of their own: a batch
                    // must not stand between a step of the chain and the next.
                    // (Round 3 measured EVERY batch as ONE 128x128 task instead of four 64x64
                    // ones, half the operand bytes per flop again, and dropped it -- N = 3072
                    // 1.29 -> 1.42 ms, N = 4096 1.90 -> 1.98: 65-us tasks on lone workgroups.
                    // Round 5: only batches of at least four steps whose tile row is 9 or more
                    // steps away (GPX_PANEL_U128 = 6 beyond the 3 of every batch) -- 125 us
                    // against 4 x 53 of workgroup time per batch of eight, value-only / with
                    // gradients N = 4096 1.34 / 2.19 -> 1.32 / 2.15 ms, N = 3072 0.95 -> 0.93,
                    // N = 2048 level; nearer batches (margin 0 / 2) cost 30 / 6 % at 4096.)
                    int s0 = s;                         // first step of this task
                    if (kbatch > 1 && q >= s + 3) {
                        if (s % kbatch != kbatch - 1 && s != q - 3) continue;
                        s0 = s - s % kbatch;
                    }
                    static const int u128 = getenv("GPX_PANEL_U128") ? atoi(getenv("GPX_PANEL_U128")) : 6;
                    const bool one = u128 >= 0 && kbatch > 1 && s - s0 + 1 >= 4 && q >= s + 3 + u128;
                    // the tile the next spine task solves (its X) in 32 x 32 tasks
                    const int fine = one ? 128 : (q == s + 1 && q < T && t == s + 1 + (stream ? 1 : 0)) ? 32 : SUB,
                              nsub = 128 / fine;
                    for (int a = 0; a < nsub; ++a)
                        for (int b = 0; b < nsub; ++b) {
                            PTask k = blank();
                            k.op = PT_GEMM_TN;
                            k.sub = (short)fine;
                            k.bufA = 0; k.offA = tile(s0, q) + fine * a;
                            k.bufB = 0; k.offB = tile(s0, t) + fine * b;
                            const int cbuf = (t > q) ? 2 : 0;      // staged / diagonal
                            const long long oc = tile(q, t) + (long long)(fine * a) * ld +
                                                 fine * b;
                            k.bufCin = (short)cbuf; k.offCin = oc;
                            k.bufCout = (short)cbuf; k.offCout = oc;
                            k.klo = 0;
                            k.khi = 128 * (s - s0 + 1);
                            k.neg = 1;
                            k.beta1 = 1;
                            dep(k, cA(s, q), r_ready(s));
                            if (t != q) dep(k, cA(s, t), r_ready(s));
                            dep(k, cA(q, t), STAGE * s0);
                            // gates, as above: 0 for extra tiles of the block's rows, 1 for
                            // the tiles of the next diagonal block
                            if (t >= T && s == 0 && !aug) dep(k, nctr() + (q >= T ? 1 : 0), 1);
                            if (one) push(k, cA(q, t), STAGE * (s - s0 + 1), 6.0 + 2.2 * (k.khi / 16));
                            else push(k, cA(q, t), (fine == SUB ? U : 1) * (s - s0 + 1),
                                      gemm_us(0, k.khi, fine));
                        }
                }
        }
    }

    // predecessors of a task from its counter thresholds: (task, through its early signal)
    void preds(int id, std::vector<std::pair<int, int>> &out) const
    {
        out.clear();
        const PTask &t = tasks[id];
        const int nd = t.ndep + t.nhost;               // (+ the tasks whose data it polls)
        for (int i = 0; i < nd; ++i) {
            const int c = t.dep[i];
            if (c >= nctr()) continue;                 // a gate: moved from outside
            for (size_t k = 0; k < signalers[c].size(); ++k) {
                const int before = k ? sigcum[c][k - 1] : 0;
                const int who = signalers[c][k];
                if (before < t.thr[i])
                    out.push_back({who, (tasks[who].sig2 == c && early[who] >= 0.0) ? 1 : 0});
            }
        }
    }

    // every predecessor of a task -- counters and polled data alike -- was generated before it
    // AND the data it follows without a counter comes from an earlier task: the leaf whose
    // panels a solve of row s takes from the mailbox (F(0), or the tail of UF(s) / XSF(s))
    bool solo_order_ok() const
    {
        std::vector<std::pair<int, int>> tmp;
        std::vector<int> leaf_at(T, -1);
        for (int i = 0; i < (int)tasks.size(); ++i) {
            const PTask &t = tasks[i];
            preds(i, tmp);
            for (const auto &pr : tmp)
                if (pr.first >= i) return false;
            if (t.op == PT_LEAF || t.op == PT_UF || (t.op == PT_XS && t.beta1 == 2))
                leaf_at[t.goff / 128] = i;
            if (t.op == PT_XS) {
                const int row = (int)((t.offA >> PT_LD_SHIFT) / 128);
                if (row < 0 || row >= T || leaf_at[row] < 0) return false;
            }
        }
        return true;
    }

    // list-scheduling simulation on `workers` workgroups, critical path first;
    // returns the start order (a topological order)
    std::vector<int> schedule(int workers) const
    {
        const int n = (int)tasks.size();
        // su[kind][i]: successors released when i finishes (0) / fires its early signal (1)
        std::vector<std::vector<int>> su[2] = {std::vector<std::vector<int>>(n),
                                               std::vector<std::vector<int>>(n)};
        std::vector<int> left(n, 0);
        std::vector<std::pair<int, int>> tmp;
        for (int i = 0; i < n; ++i) {
            preds(i, tmp);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            for (size_t k = 0; k < tmp.size(); ++k) {
                // both kinds from one task: the later (finish) one decides
                if (tmp[k].second == 1 && k > 0 && tmp[k - 1].first == tmp[k].first) continue;
                if (tmp[k].second == 0 && k + 1 < tmp.size() && tmp[k + 1].first == tmp[k].first) {
                    su[0][tmp[k].first].push_back(i);
                    ++left[i];
                    ++k;
                    continue;
                }
                su[tmp[k].second][tmp[k].first].push_back(i);
                ++left[i];
            }
        }
        std::vector<double> level(n, 0.0);            // longest path to the end
        for (int i = n - 1; i >= 0; --i) {            // generation order is topological
            double m = 0.0;
            for (int q : su[0][i]) m = std::max(m, cost[i] + level[q]);
            for (int q : su[1][i]) m = std::max(m, early[i] + level[q]);
            level[i] = std::max(m, cost[i]);
        }
        std::vector<int> order;
        std::vector<double> wfree(workers, 0.0);
        // ready tasks, highest level first, lowest index among equals (a heap: the graph of
        // a whole 4096-matrix has 25 000 tasks, the linear search was quadratic)
        std::priority_queue<std::pair<double, int>> ready;
        for (int i = 0; i < n; ++i)
            if (left[i] == 0) ready.push({level[i], -i});
        struct Event {
            double at;
            int task, kind;
            long long seq;                            // equal times: first in, first out
        };
        auto later = [](const Event &a, const Event &b) {
            return a.at > b.at || (a.at == b.at && a.seq > b.seq);
        };
        std::priority_queue<Event, std::vector<Event>, decltype(later)> running(later);
        long long seq = 0;
        double now = 0.0;
        while ((int)order.size() < n) {
            // workers free at `now` take ready tasks
            while (!ready.empty()) {
                int w = -1;
                for (int k = 0; k < workers; ++k)
                    if (wfree[k] <= now) { w = k; break; }
                if (w < 0) break;
                const int best = -ready.top().second;
                ready.pop();
                order.push_back(best);
                wfree[w] = now + cost[best];
                running.push({wfree[w], best, 0, seq++});
                if (!su[1][best].empty()) running.push({now + early[best], best, 1, seq++});
            }
            if ((int)order.size() == n) break;
            // advance to the next event
            if (running.empty()) return {};           // cannot happen: graph is acyclic
            const Event ev = running.top();
            running.pop();
            now = std::max(now, ev.at);
            for (int q : su[ev.kind][ev.task])
                if (--left[q] == 0) ready.push({level[q], -q});
        }
        return order;
    }
};

int env_once(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// whole-matrix launches batch the early updates of far tiles (GPX_PANEL_KBATCH, default 4)
int panel_kbatch(int T, int E)
{
    // (measured, value-only evaluation, with the two updates before a tile's row unbatched:
    // 2 / 3 / 4 / 6 / 8 / 10 / 12 / 16 steps per task give N = 4096 2.03 / 1.91 / 1.90 /
    // 1.80 / 1.77 / 1.79 / 1.78 / 1.83 ms, N = 3072 1.31 / 1.30 / 1.28 / 1.24 / 1.23 / 1.26 /
    // 1.26 / 1.31, N = 2048 0.78 / 0.77 / 0.79 / 0.77 / 0.77 / 0.79 / 0.81 / 0.82; 0.79 / 1.58 /
    // 2.69 without batching)
    static const int kb = [] {
        const int v = env_once("GPX_PANEL_KBATCH", -1);
        return v < 1 ? -1 : (v > 16 ? 16 : v);
    }();
    if (T <= GPX_PANEL_MAX / 128 || E > 1) return 1;
    return kb > 0 ? kb : 8;
}

// the diagonal update and leaf of a tile on a workgroup of their own that follows the spine's
// solve (round 5, PT_UF; GPX_PANEL_SPLIT=0: the fused spine task of rounds 2-4). Not for wide
// panels and not for the round-1 graph.
bool panel_split(bool stream, int E, bool aug)
{
    static const int on = env_once("GPX_PANEL_SPLIT", 1);
    return on && stream && (E == 0 || aug);
}

// (with the split spine) the spine's solves fold the last update of their tile in themselves
// (GPX_PANEL_FOLD=0: sixteen worker products apply it)
bool panel_fold()
{
    static const int on = env_once("GPX_PANEL_FOLD", 1);
    return on != 0;
}

struct PanelList {
    PTask *dev = nullptr;                    // [ntasks] general tasks, then [nspine] leaves
    int ntasks = 0, nspine = 0, nctr = 0;
};

// solo (round 5): the list of a launch whose workgroups each run ONE member's whole graph, task
// after task in generation order (a topological order that also holds for the tasks whose data
// a follower polls: every producer has finished before its consumer starts) -- all tasks are
// "spine" tasks of a single spine workgroup per member, the queue is empty. solo = 2: without
// the tasks of the inverse (nothing reads W: value-only members).
int panel_list(int T, int E, int workers, bool aug, int ig, PanelList *out, int solo = 0)
{
    typedef std::tuple<int, int, int, int, int, int> Key;
    static std::map<Key, PanelList> cache;
    static std::mutex mu;
    int device = 0;
    GPX_HIP(hipGetDevice(&device));
    static const int stream = env_once("GPX_PANEL_STREAM", 1);
    const Key key(device, T, aug ? -1 : E, solo ? 0 : workers, ig, solo);   // (aug: E = 1, a right-hand side)
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return 0;
    }
    // (round 4: the lists no longer depend on the leading dimension -- pt_off -- so the
    // cache holds one entry per graph shape and worker count, a small fixed set)
    Graph g;
    g.T = T;
    g.E = E;
    g.ld = PT_LD;
    g.stream = stream != 0;
    g.kbatch = panel_kbatch(T, E);
    g.aug = aug;
    g.split = panel_split(stream != 0, E, aug);
    g.fold = panel_fold();
    g.ig = ig;
    g.build();
    std::vector<PTask> sorted, leaves;
    if (solo) {
        if (!g.solo_order_ok()) {
            gpx_set_error("panel: generation order is not a sequential order (T = %d)", T);
            return -1;
        }
        for (const PTask &t : g.tasks)
            if (solo != 2 || t.sig < g.cX(0, 0)) leaves.push_back(t);   // (cX, cW: the inverse)
    } else {
    const std::vector<int> order = g.schedule(workers);
    if (order.size() != g.tasks.size()) {
        gpx_set_error("panel: scheduling failed (T = %d, E = %d)", T, E);
        return -1;
    }
    sorted.reserve(order.size());
    for (int id : order) {
        const PTask &t = g.tasks[id];
        const bool chain = t.op == PT_LEAF || t.op == PT_UF ||
                           (t.op == PT_XS && (t.beta1 == 2 || t.spine));
        (chain ? leaves : sorted).push_back(t);
    }
    // the spine walks the diagonal in order (the simulation may start XSF(1), which has
    // no counter to wait for, ahead of F(0), whose panels it follows)
    std::stable_sort(leaves.begin(), leaves.end(),
                     [](const PTask &a, const PTask &b) { return a.goff < b.goff; });
    }
    PanelList pl;
    pl.ntasks = (int)sorted.size();
    pl.nspine = (int)leaves.size();
    sorted.insert(sorted.end(), leaves.begin(), leaves.end());
    pl.nctr = g.nctr();
    GPX_HIP(hipMalloc((void **)&pl.dev, sorted.size() * sizeof(PTask)));
    GPX_HIP(hipMemcpy(pl.dev, sorted.data(), sorted.size() * sizeof(PTask),
                      hipMemcpyHostToDevice));
    GPX_HIP(hipDeviceSynchronize());      // in HBM before any stream reads it
    cache[key] = pl;
    *out = pl;
    return 0;
}

}  // namespace

// Host-side self-check of the task graph of one panel launch (no GPU needed): the
// schedule is a permutation and a topological order of the counter dependencies (every
// threshold a task waits for is reached by signals of tasks before it, an early signal
// counting from its task on), every counter ends where the graph says a finished tile
// stands, and the spine has one task per diagonal tile. Returns 0, or -1 with
// gpx_last_error() naming the first violation.
static int panel_graph_check(int T, int E, int workers, int stream, int *ntasks, bool aug = false,
                             bool full_w = false);
// Co-residency of a (member-batched) launch: each of its workgroups holds a whole CU, the
// spine workgroups come first in the grid, and the progress argument needs all of them plus
// at least one worker resident together.
static bool gpx_panel_grid_fits(int nmem, int nspwg, int ncu)
{
    return nmem >= 1 && nspwg >= 1 && (long long)nmem * nspwg + 1 <= ncu;
}
// spine workgroups per member by the default rule of gpx_panel (3 up to 16 members, 2 up to
// 40, 1 beyond), reduced until the launch fits a device of ncu CUs; -1: it cannot fit
extern "C" int gpx_panel_grid_check(int nmem, int ncu, int *nspwg, int *workers)
{
    if (nmem < 1 || ncu < 1) {
        gpx_set_error("panel grid check: bad arguments");
        return -1;
    }
    int sp = nmem == 1 ? 3 : (nmem <= 16 ? 3 : (nmem <= 40 ? 2 : 1));
    while (sp > 1 && !gpx_panel_grid_fits(nmem, sp, ncu)) --sp;
    if (!gpx_panel_grid_fits(nmem, sp, ncu)) {
        gpx_set_error("panel grid check: %d members do not fit %d CUs", nmem, ncu);
        return -1;
    }
    if (nspwg) *nspwg = sp;
    if (workers) *workers = std::max(8, 250 - nmem * sp);
    return 0;
}
extern "C" int gpx_panel_graph_check(int T, int workers, int stream, int *ntasks)
{
    return panel_graph_check(T, 0, workers, stream, ntasks);
}
// the same with one more tile column right of the block that covers a WHOLE matrix of T
// tiles (2..32): the right-hand side of the forward substitution (round 4: also for the
// matrices of at most 8 tiles that are one panel)
extern "C" int gpx_panel_graph_check_rhs(int T, int workers, int *ntasks)
{
    return panel_graph_check(T, 1, workers, 1, ntasks, true);
}
// ... and with the WHOLE inverse assembled in the launch (the route of an evaluation with
// gradients up to 32 tiles, round 5)
extern "C" int gpx_panel_graph_check_full(int T, int workers, int *ntasks)
{
    return panel_graph_check(T, 1, workers, 1, ntasks, true, true);
}
// Solo launches (one workgroup per member runs the member's whole graph, gpx_panel_solo): the
// list is the graph in GENERATION order. Host check that this is a sequential order for T
// tiles [+ a right-hand side; full: with the whole inverse; value_only: without the tasks of
// the inverse]: every task finds every counter it waits for reached by tasks before it, every
// follower its producers finished, every solve the leaf of its row, and the counters end where
// a finished tile stands. *ntasks = tasks of the list.
extern "C" int gpx_panel_solo_check(int T, int aug, int full, int value_only, int *ntasks)
{
    if (T < 2 || T > PCTL_TMAX) {
        gpx_set_error("panel solo check: bad arguments");
        return -1;
    }
    Graph g;
    g.T = T;
    g.E = aug ? 1 : 0;
    g.ld = PT_LD;
    g.stream = true;
    g.kbatch = panel_kbatch(T, g.E);
    g.aug = aug != 0;
    g.split = panel_split(true, g.E, g.aug);
    g.fold = panel_fold();
    g.ig = full && T > PANEL_IG ? T : PANEL_IG;
    g.build();
    if (!g.solo_order_ok()) {
        gpx_set_error("panel solo check: generation order is not sequential (T = %d)", T);
        return -1;
    }
    std::vector<int> ctr(g.nctr(), 0);
    int count = 0;
    for (size_t id = 0; id < g.tasks.size(); ++id) {
        const PTask &t = g.tasks[id];
        if (value_only && t.sig >= g.cX(0, 0)) continue;   // (a task of the inverse)
        for (int i = 0; i < t.ndep + t.nhost; ++i) {
            if (t.dep[i] >= g.nctr()) continue;
            if (ctr[t.dep[i]] < t.thr[i]) {
                gpx_set_error("panel solo check: task %d (op %d) needs counter %d >= %d, which "
                              "stands at %d", (int)id, t.op, (int)t.dep[i], (int)t.thr[i],
                              ctr[t.dep[i]]);
                return -1;
            }
        }
        ctr[t.sig] += t.siginc;
        if (t.sig2 >= 0) ctr[t.sig2] += Graph::STAGE;
        ++count;
    }
    for (int s2 = 0; s2 < T; ++s2)
        for (int t = s2; t < g.TW(); ++t)
            if (ctr[g.cA(s2, t)] != Graph::STAGE * (s2 + 1)) {
                gpx_set_error("panel solo check: tile (%d,%d) ends at %d, not %d", s2, t,
                              ctr[g.cA(s2, t)], Graph::STAGE * (s2 + 1));
                return -1;
            }
    if (ntasks) *ntasks = count;
    return 0;
}
// the same for a wide panel: E more tile columns right of the block (row panel and update
// of the next diagonal block inside the launch); round-2 graph only
extern "C" int gpx_panel_graph_check_wide(int T, int E, int workers, int *ntasks)
{
    // (E = 1 right of a matrix of more than 8 tiles has always been a right-hand side)
    return panel_graph_check(T, E, workers, 1, ntasks, T > GPX_PANEL_MAX / 128 && E == 1);
}
static int panel_graph_check(int T, int E, int workers, int stream, int *ntasks, bool aug,
                             bool full_w)
{
    if (T < 2 || T > PCTL_TMAX || workers < 1 || E < 0 || E > GPX_PANEL_MAX / 128 ||
        (E > 0 && !stream) || (T > GPX_PANEL_MAX / 128 && (!stream || E > 1))) {
        gpx_set_error("panel graph check: bad arguments");
        return -1;
    }
    Graph g;
    g.T = T;
    g.E = E;
    g.ld = 128 * (T + E);
    g.stream = stream != 0;
    g.kbatch = panel_kbatch(T, E);
    g.aug = aug;
    g.split = panel_split(stream != 0, E, aug);
    g.fold = panel_fold();
    g.ig = full_w && T > PANEL_IG ? T : PANEL_IG;
    g.build();
    const int n = (int)g.tasks.size();
    if (ntasks) *ntasks = n;
    const std::vector<int> order = g.schedule(workers);
    if ((int)order.size() != n) {
        gpx_set_error("panel graph check: schedule has %d of %d tasks", (int)order.size(), n);
        return -1;
    }
    std::vector<char> seen(n, 0);
    std::vector<int> ctr(g.nctr(), 0);
    unsigned long long last_spine = 0;                   // bit per diagonal tile
    for (int pos = 0; pos < n; ++pos) {
        const int id = order[pos];
        if (id < 0 || id >= n || seen[id]) {
            gpx_set_error("panel graph check: position %d repeats or invents task %d", pos, id);
            return -1;
        }
        seen[id] = 1;
        const PTask &t = g.tasks[id];
        for (int i = 0; i < t.ndep + t.nhost; ++i)
            if (t.dep[i] < g.nctr() && ctr[t.dep[i]] < t.thr[i]) {
                gpx_set_error("panel graph check: task %d (op %d) at position %d waits for "
                              "counter %d >= %d, which stands at %d", id, t.op, pos,
                              (int)t.dep[i], (int)t.thr[i], ctr[t.dep[i]]);
                return -1;
            }
        ctr[t.sig] += t.siginc;
        if (t.sig2 >= 0) ctr[t.sig2] += Graph::STAGE;
        if (t.op == PT_LEAF || t.op == PT_UF || (t.op == PT_XS && t.beta1 == 2)) {
            // one spine task per diagonal tile; a fused one follows the tile before it
            const int tile = (int)(t.goff / 128);
            if (tile < 0 || tile >= T || (last_spine >> tile & 1)) {
                gpx_set_error("panel graph check: spine task of tile %d twice or out of range", tile);
                return -1;
            }
            last_spine |= 1ull << tile;
        }
    }
    if (last_spine != (1ull << T) - 1) {
        gpx_set_error("panel graph check: spine tiles %#llx of %d", last_spine, T);
        return -1;
    }
    for (int s = 0; s < T + E; ++s)
        for (int t = s; t < T + E; ++t) {
            // a tile of the block's rows: s updates and its row-panel step; a tile of the
            // next diagonal block: the T updates of this block
            const int want = (g.aug && s >= T) ? 0 : Graph::STAGE * (s < T ? s + 1 : T);
            if (ctr[g.cA(s, t)] != want) {
                gpx_set_error("panel graph check: tile (%d,%d) ends at %d, not %d", s, t,
                              ctr[g.cA(s, t)], want);
                return -1;
            }
            if (t >= T || s / g.ig != t / g.ig) continue;   // W: inside 1024-blocks (or all of it)
            if (t > s && (ctr[g.cX(s, t)] != Graph::STAGE * g.inv_chunks(s, t) ||
                          ctr[g.cW(s, t)] != Graph::STAGE)) {
                gpx_set_error("panel graph check: inverse tile (%d,%d) incomplete (%d, %d)", s, t,
                              ctr[g.cX(s, t)], ctr[g.cW(s, t)]);
                return -1;
            }
        }
    return 0;
}

// ---- host: phases of a lock-step sweep ----------------------------------------------
namespace {

struct SweepList {
    PTask *dev = nullptr;
    std::vector<int> first, count;           // per phase: 0 = F(0), 1 + s = X(s)
    int *ctl = nullptr;                      // constant control block
};

int sweep_list(int T, int E, SweepList **out)
{
    typedef std::tuple<int, int, int> Key;
    static std::map<Key, SweepList> cache;
    static std::mutex mu;
    int device = 0;
    GPX_HIP(hipGetDevice(&device));
    const Key key(device, T, E);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = &it->second;
        return 0;
    }
    Graph g;
    g.T = T;
    g.E = E;
    g.ld = PT_LD;
    g.stream = true;
    g.kbatch = 1;
    g.aug = E == 1;
    g.build();
    SweepList sl;
    std::vector<PTask> all;
    sl.first.assign(T + 1, 0);
    sl.count.assign(T + 1, 0);
    auto strip = [](PTask t) {
        t.ndep = 0;
        t.klo = 0;                           // counter 0 of the constant block: published
        t.sig2 = -1;                         // no early signal
        if (t.op == PT_XS) t.bufA = 0;       // R_ss and W_ss are final: read them themselves
        if (t.op == PT_LEAF) t.khi = 0;      // no streaming
        return t;
    };
    for (int ph = 0; ph <= T; ++ph) {
        sl.first[ph] = (int)all.size();
        for (int pass = 0; pass < 2; ++pass)            // fused tasks first
            for (const PTask &t : g.tasks) {
                const bool fused = t.op == PT_XS && t.beta1 == 2;
                if (ph == 0) {
                    if (pass == 0 && t.op == PT_LEAF && t.offA == 0) all.push_back(strip(t));
                    continue;
                }
                const int s = ph - 1;
                if (t.op != PT_XS || t.offA != g.tile(s, s) || fused != (pass == 0)) continue;
                all.push_back(strip(t));
            }
        sl.count[ph] = (int)all.size() - sl.first[ph];
    }
    GPX_HIP(hipMalloc((void **)&sl.dev, all.size() * sizeof(PTask)));
    GPX_HIP(hipMemcpy(sl.dev, all.data(), all.size() * sizeof(PTask), hipMemcpyHostToDevice));
    int hctl[PCTL_HEAD + 4] = {};
    hctl[PCTL_HEAD] = 1 << 30;
    GPX_HIP(hipMalloc((void **)&sl.ctl, sizeof(hctl)));
    GPX_HIP(hipMemcpy(sl.ctl, hctl, sizeof(hctl), hipMemcpyHostToDevice));
    GPX_HIP(hipDeviceSynchronize());
    cache[key] = sl;
    *out = &cache[key];
    return 0;
}

}  // namespace

// phase 0: the leaf of tile (0,0); phase 1 + s: the row panel of tile row s -- XS(s,t) for
// every tile right of the diagonal, the one next to it fused with the update and the leaf
// of tile (s+1,s+1) -- for the block (off, 128 T) [+ `aug` right-hand-side column] of every
// member of the workspace. The caller applies the trailing updates between the phases.
int gpx_sweep_phase(hipStream_t s, const DenseWs &w, int off, int T, bool aug, int phase,
                    bool no_inverse, bool fused_only, bool presolved)
{
    if (T < 1 || T > PCTL_TMAX || phase < 0 || phase > T) {
        gpx_set_error("sweep: bad phase %d of %d tiles", phase, T);
        return -1;
    }
    SweepList *sl = nullptr;
    GPX_TRY(sweep_list(T, aug ? 1 : 0, &sl));
    // (fused_only: the row-panel tiles go to gpx_sweep_xs; what is left of phase 1 + s is the
    // fused task of tile (s, s+1), the first of the phase's list -- none in the last phase)
    const int ntask = fused_only ? (phase >= 1 && phase < T ? 1 : 0) : sl->count[phase];
    if (ntask == 0) return 0;
    GPX_TRY(gpx_test_jitter(s));
    static const int timeout_ms = [] {
        const int v = env_once("GPX_PANEL_TIMEOUT_MS", 2000);
        return v < 1 ? 2000 : v;
    }();
    static const int strict = env_once("GPX_PANEL_STRICT", 0);
    static const int leafskip = env_once("GPX_PANEL_LEAF_SKIP", 0) |
                                (env_once("GPX_LEAF_MFMA", 1) ? 0 : 32);
    const size_t o = (size_t)off * w.ld + off;
    const int nmem = w.batch > 1 ? w.batch : 1;
    SweepArgs p;
    p.bA = w.A + o;
    p.bW = w.W + o;
    p.bX = w.Kinv + o;
    p.ld = w.ld;
    p.mstride = nmem > 1 ? w.mstride : 0;
    p.tasks = sl->dev + sl->first[phase];
    p.ntasks = ntask;
    p.nmem = nmem;
    p.ctl = sl->ctl;
    p.info = w.info;
    p.goff = off;
    p.timeout = (long long)timeout_ms * 100000LL;
    p.strict = strict;
    // no_inverse: nothing will read W beyond the diagonal 16-blocks the solves use
    p.leafskip = leafskip | (no_inverse ? 8 : 0);
    p.presolved = fused_only && presolved ? 1 : 0;
    hipLaunchKernelGGL(sweep_kernel, dim3(p.ntasks * nmem), dim3(256), LEAF2_LDS, s, p);
    GPX_HIP(hipGetLastError());
    return 0;
}

// The row-panel tiles (s, t), t = t0 .. TW-1, of every member as one dense launch
// (sweep_xs_kernel): each applies the trailing updates kfirst .. s-1 of its tile and solves it.
int gpx_sweep_xs(hipStream_t st, const DenseWs &w, int off, int T, bool aug, int s, int t0,
                 int kfirst, int upd, int kfirst_rhs)
{
    const int TW = T + (aug ? 1 : 0);
    if (T < 1 || T > PCTL_TMAX || s < 0 || s >= T || t0 <= s || t0 > TW || kfirst < 0 || kfirst > s) {
        gpx_set_error("sweep: bad row panel (%d, %d.., from step %d) of %d tiles", s, t0, kfirst, T);
        return -1;
    }
    // upd = 2: tile (s, s+1) and the diagonal tile (s+1, s+1) take their steps kfirst .. s-1 in
    // this launch as well (update only; sweep_kernel's fused task solves / factors them);
    // upd = 1: the diagonal tile only (tile (s, s+1) is one of the solved tiles: t0 = s + 1);
    // upd = 3 (right-looking): EVERY tile of the rows below s takes the steps kfirst .. s-1
    if (upd < 0 || upd > 3 || ((upd == 1 || upd == 3) && t0 != s + 1) ||
        (upd == 2 && s + 1 < T && t0 != s + 2)) {
        gpx_set_error("sweep: bad update roles (%d) for tile row %d from column %d", upd, s, t0);
        return -1;
    }
    int extra = upd && upd < 3 && s + 1 < T && kfirst < s ? upd : 0;
    if (upd == 3 && kfirst < s)
        for (int u = s + 1; u < T; ++u) extra += TW - u;
    if (t0 == TW && !extra) return 0;
    GPX_TRY(gpx_test_jitter(st));
    const size_t o = (size_t)off * w.ld + off;
    const int nmem = w.batch > 1 ? w.batch : 1;
    SweepXsArgs p;
    p.bA = w.A + o;
    p.bW = w.W + o;
    p.bX = w.Kinv + o;
    p.ld = w.ld;
    p.mstride = nmem > 1 ? w.mstride : 0;
    p.nmem = nmem;
    p.s = s;
    p.t0 = t0;
    p.kfirst = kfirst;
    p.nsolve = TW - t0;
    p.TW = TW;
    // (GPX_SWEEP_NARROW=0: the right-hand-side tiles as full tiles)
    static const int narrow_on = env_once("GPX_SWEEP_NARROW", 1);
    p.narrow_col = aug && narrow_on ? T : -1;
    // (the narrow tiles may fold from an earlier step on than the others: their updates cost an
    // eighth, and the products of the tile engine that would apply them run full tiles)
    p.kfirst_narrow = p.narrow_col >= 0 && kfirst_rhs >= 0 && kfirst_rhs <= kfirst ? kfirst_rhs : kfirst;
    p.right = upd == 3 ? 1 : 0;
    p.trace = nullptr;
    static const int debug = env_once("GPX_XS_DEBUG", 0);    // developer aid: stamps of workgroup 0
    static long long *trace_dev = nullptr;
    if (debug) {
        if (!trace_dev) GPX_HIP(hipMalloc((void **)&trace_dev, 32 * sizeof(long long)));
        GPX_HIP(hipMemsetAsync(trace_dev, 0, 32 * sizeof(long long), st));
        p.trace = trace_dev;
    }
    hipLaunchKernelGGL(sweep_xs_kernel, dim3((TW - t0 + extra) * nmem), dim3(256), XSL_LDS, st, p);
    GPX_HIP(hipGetLastError());
    if (debug) {
        long long h[32];
        GPX_HIP(hipStreamSynchronize(st));
        GPX_HIP(hipMemcpy(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "xs trace s=%d t0=%d kf=%d wgs=%d (us from start): tile-in %.2f updates %.2f "
                "panels %.2f last-step %.2f stores %.2f | steps", s, t0, kfirst, (TW - t0) * nmem,
                (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, (h[3] - h[0]) * 0.01,
                (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %.2f", (h[8 + i] - h[0]) * 0.01);
        fprintf(stderr, "\n");
    }
    return 0;
}

// Host-side self-check of the lock-step sweep (no GPU needed): replays the phases in the
// order sweep_block (chol.hip) issues them -- F(0); for every tile row s the left-looking
// updates of the tile engine (row s from steps 0 .. s-1, the next diagonal tile from steps
// 0 .. s-1), then the row-panel phase X(s) -- against the counter thresholds of the panel
// launch's own task graph: every task of a phase finds the counters it would have waited
// for already there, every product finds the row panels it multiplies final, every tile
// ends where the graph says a finished tile stands, and the phases hold every leaf / XS
// task of the graph exactly once (fused tasks first). Returns 0, or -1 with
// gpx_last_error() naming the first violation.
// Trailing updates a dense row-panel task applies itself (the last `depth` steps before its
// row; earlier ones stay one product of the tile engine): ALL of them. With four launches a
// tile row the depth did not matter; with two, narrow right-hand-side tiles and 128 members a
// group the dense tasks beat the products of the tile engine at every depth measured (256
// thetas value-only, depth 4 / 8 / 16 / 32: N = 2048 14.5k / 15.1k / 15.4k, N = 4096 2.17k /
// 2.20k / 2.24k / 2.29k evals/s; N = 3072 level from 16 on). GPX_SWEEP_FOLD overrides.
int gpx_sweep_fold_depth(int T)
{
    static const int depth_env = env_once("GPX_SWEEP_FOLD", -1);
    return depth_env >= 0 ? depth_env : T;
}
bool gpx_sweep_lite()
{
    static const int lite = env_once("GPX_SWEEP_LITE", 1);
    return lite != 0;
}

static int sweep_check_impl(int T, int aug, bool lite, int depth, bool right = false)
{
    if (T < 1 || T > PCTL_TMAX) {
        gpx_set_error("sweep check: bad arguments");
        return -1;
    }
    Graph g;
    g.T = T;
    g.E = aug ? 1 : 0;
    g.ld = PT_LD;
    g.stream = true;
    g.kbatch = 1;
    g.aug = aug != 0;
    g.build();
    const int TW = g.TW(), STAGE = Graph::STAGE;
    std::vector<int> ctr(g.nctr(), 0);
    std::vector<char> used(g.tasks.size(), 0);
    auto run_task = [&](int id, int phase) -> int {
        const PTask &t = g.tasks[id];
        if (used[id]) {
            gpx_set_error("sweep check: task %d in two phases", id);
            return -1;
        }
        used[id] = 1;
        for (int i = 0; i < t.ndep; ++i) {
            if (t.dep[i] >= g.nctr()) continue;
            if (ctr[t.dep[i]] < t.thr[i]) {
                gpx_set_error("sweep check: phase %d, task %d (op %d) needs counter %d >= %d, "
                              "which stands at %d", phase, id, t.op, (int)t.dep[i], (int)t.thr[i],
                              ctr[t.dep[i]]);
                return -1;
            }
        }
        ctr[t.sig] += t.siginc;
        if (t.sig2 >= 0) ctr[t.sig2] += STAGE;
        return 0;
    };
    // phase 0: the leaf of tile (0, 0)
    for (size_t id = 0; id < g.tasks.size(); ++id)
        if (g.tasks[id].op == PT_LEAF && g.tasks[id].offA == 0) GPX_TRY(run_task((int)id, 0));
    for (int s = 0; s < T; ++s) {
        // dense row panels (sweep_block with GPX_SWEEP_LITE): every tile right of the diagonal
        // and the next diagonal tile take the steps before kf as one product of the tile engine
        // and the steps kf .. s-1 inside the dense launch, whose tasks then solve the tiles from
        // t0 on (tile (s, s+1) and the diagonal tile stay with the fused task)
        const int t0 = lite ? (s + 1 < T ? s + 2 : s + 1) : TW,
                  kf = right ? std::max(0, s - 1) : (lite ? std::max(0, s - depth) : s);
        if (right && s >= 1) {
            // right-looking: the dense launch of row s applies step s-1 to EVERY tile from row
            // s down (the tiles of row s then have all their steps; no products at all)
            for (int u = s; u < T; ++u)
                for (int t = u == s ? s + 1 : u; t < TW; ++t) {
                    if (ctr[g.cA(s - 1, u)] < Graph::r_ready(s - 1) ||
                        ctr[g.cA(s - 1, t)] < Graph::r_ready(s - 1)) {
                        gpx_set_error("sweep check: step %d of tile (%d,%d) before its rows are final",
                                      s - 1, u, t);
                        return -1;
                    }
                    if (ctr[g.cA(u, t)] != STAGE * (s - 1)) {
                        gpx_set_error("sweep check: tile (%d,%d) at %d, not %d, when step %d comes", u,
                                      t, ctr[g.cA(u, t)], STAGE * (s - 1), s - 1);
                        return -1;
                    }
                    ctr[g.cA(u, t)] += STAGE;
                }
        }
        auto rows_final = [&](int j0, int j1, int c0, int c1, const char *what, int ti, int tj) -> int {
            for (int j = j0; j < j1; ++j)
                if (ctr[g.cA(j, c0)] < Graph::r_ready(j) || ctr[g.cA(j, c1)] < Graph::r_ready(j)) {
                    gpx_set_error("sweep check: %s of tile (%d,%d) before R(%d,%d) / R(%d,%d) is final",
                                  what, ti, tj, j, c0, j, c1);
                    return -1;
                }
            return 0;
        };
        if (s >= 1 && !right) {
            // the tile engine: steps 0 .. kf-1 (all of them without the dense launch) ...
            // (the tile of a right-hand side takes ALL its steps in the dense launch: narrow)
            auto kf_of = [&](int t) { return lite && aug && t == TW - 1 ? 0 : kf; };
            for (int t = s + 1; t < TW; ++t) {
                GPX_TRY(rows_final(0, kf_of(t), s, t, "product", s, t));
                if (ctr[g.cA(s, t)] != 0) {
                    gpx_set_error("sweep check: tile (%d,%d) updated twice", s, t);
                    return -1;
                }
                ctr[g.cA(s, t)] += STAGE * kf_of(t);
            }
            // ... and the next diagonal tile (XSF adds step s)
            if (s + 1 < T) {
                GPX_TRY(rows_final(0, kf, s + 1, s + 1, "product", s + 1, s + 1));
                if (ctr[g.cA(s + 1, s + 1)] != 0) {
                    gpx_set_error("sweep check: diagonal tile %d updated twice", s + 1);
                    return -1;
                }
                ctr[g.cA(s + 1, s + 1)] += STAGE * kf;
            }
            // the dense launch: steps kf .. s-1
            {
                for (int t = s + 1; t < TW; ++t) {
                    const int kt = kf_of(t);
                    if (kt >= s) continue;
                    GPX_TRY(rows_final(kt, s, s, t, "update in the dense launch", s, t));
                    if (ctr[g.cA(s, t)] != STAGE * kt) {
                        gpx_set_error("sweep check: dense task finds tile (%d,%d) at %d, not %d", s, t,
                                      ctr[g.cA(s, t)], STAGE * kt);
                        return -1;
                    }
                    ctr[g.cA(s, t)] += STAGE * (s - kt);
                }
                if (kf >= s) goto dense_done;
                if (s + 1 < T) {
                    GPX_TRY(rows_final(kf, s, s + 1, s + 1, "update in the dense launch", s + 1, s + 1));
                    ctr[g.cA(s + 1, s + 1)] += STAGE * (s - kf);
                }
            dense_done:;
            }
        }
        // (the solves of the dense launch need the leaf of their row)
        if (lite && t0 < TW && ctr[g.cA(s, s)] < STAGE * (s + 1)) {
            gpx_set_error("sweep check: dense solves of row %d before its leaf", s);
            return -1;
        }
        // X(s): every XS task of row s, the fused one first (it signals R(s,s+1) early,
        // which the plain ones of the NEXT row wait for -- inside a phase nothing waits);
        // with the dense launch: its tiles first, the fused task in a launch of its own
        for (int pass = lite ? 1 : 0; pass >= 0 && pass < 2; pass += lite ? -1 : 1)
            for (size_t id = 0; id < g.tasks.size(); ++id) {
                const PTask &t = g.tasks[id];
                const bool fused = t.op == PT_XS && t.beta1 == 2;
                if (t.op != PT_XS || t.offA != g.tile(s, s) || fused != (pass == 0)) continue;
                GPX_TRY(run_task((int)id, 1 + s));
            }
    }
    for (size_t id = 0; id < g.tasks.size(); ++id) {
        const PTask &t = g.tasks[id];
        if ((t.op == PT_LEAF || t.op == PT_XS) && !used[id]) {
            gpx_set_error("sweep check: task %d (op %d) is in no phase", (int)id, t.op);
            return -1;
        }
    }
    for (int s = 0; s < T; ++s)
        for (int t = s; t < TW; ++t)
            if (ctr[g.cA(s, t)] != STAGE * (s + 1)) {
                gpx_set_error("sweep check: tile (%d,%d) ends at %d, not %d", s, t, ctr[g.cA(s, t)],
                              STAGE * (s + 1));
                return -1;
            }
    return 0;
}

extern "C" int gpx_sweep_check(int T, int aug) { return sweep_check_impl(T, aug, false, 0); }
// ... with the dense row panels (sweep_xs_kernel); depth < 0: the library's rule for T tiles
extern "C" int gpx_sweep_check_lite(int T, int aug, int depth)
{
    // depth = -2: the right-looking form (small matrices: every launch applies one step to all
    // tiles below its row)
    if (depth == -2) return sweep_check_impl(T, aug, true, 0, true);
    return sweep_check_impl(T, aug, true, depth < 0 ? gpx_sweep_fold_depth(T) : depth);
}
// the right-hand-side tiles of a sweep are narrow (sweep_xs_kernel): the caller may leave ALL
// their updates to the dense tasks
bool gpx_sweep_narrow()
{
    static const int narrow_on = env_once("GPX_SWEEP_NARROW", 1);
    return narrow_on != 0;
}
// tiles up to which the sweep is right-looking (GPX_SWEEP_RIGHT; 0: never)
int gpx_sweep_right_max()
{
    static const int v = env_once("GPX_SWEEP_RIGHT", 4);
    return v;
}

int gpx_panel_init()
{
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LEAF2_LDS));
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LEAF2_LDS));
    GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_xs_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, XSL_LDS));
    return 0;
}

// largest diagonal block handled by one panel launch for a matrix of padded order
// np (0: recursion down to the leaves). One panel launch stands for ~36 dependent
// launches of the recursion; with its tile hand-offs as 16-B sc1 accesses a 1024-block
// takes 0.54 ms against 0.66 ms, and it is the default at every size (N = 16384:
// value-only evaluations 31.0 -> 29.3 ms, batches +0.7%). GPX_PANEL=0 / 256..1024
// overrides.
int gpx_panel_max(int np)
{
    // magic statics: the host threads of the multi-device path reach this together
    static const int forced = [] {
        const char *e = getenv("GPX_PANEL");
        int f = e ? atoi(e) : -1;
        if (f > 0 && (f < 256 || f > GPX_PANEL_MAX || f % 128)) f = GPX_PANEL_MAX;
        return f;
    }();
    (void)np;
    return forced >= 0 ? forced : GPX_PANEL_MAX;
}

bool gpx_panel_streaming()
{
    static const int stream = env_once("GPX_PANEL_STREAM", 1);
    return stream != 0;
}

// Solo launches (round 5): the diagonal blocks of a group of MANY members at a SMALL order as
// one launch with one workgroup per member that runs the member's whole task graph, task after
// task (panel_list, solo). The lock-step sweep of such a group is bound by occupancy, not by
// arithmetic: every tile task holds a whole CU (157 KB of LDS) for 26-72 us while it uses the
// matrix pipe for 8-20, a phase of a few hundred of them fills the device whatever else could
// run, and the rank-128 .. 384 products between the phases move more bytes than they compute
// on (N = 512: 128 members 0.50 ms, two groups side by side 0.99 ms -- 13 % of overlap).
// One workgroup per member needs no hand-off at all -- every poll finds its data -- keeps the
// tile it solves in registers while it folds the update before it in (xs_run, uf_run: the
// split graph's bodies), and 128 members are 128 CUs: the other group in flight has the other
// half of the device. Same task bodies, same order per element: the bits of the panel launch
// and of the sweep. GPX_SOLO_MAX_NP (largest padded order, 0: never), GPX_SOLO_MIN_MEMBERS.
bool gpx_panel_solo_np(int np, int members)
{
    static const int max_np = env_once("GPX_SOLO_MAX_NP", 0);
    static const int min_members = env_once("GPX_SOLO_MIN_MEMBERS", 16);
    return min_members > 0 && members >= min_members && np >= 256 && np <= max_np &&
           gpx_panel_streaming();
}
bool gpx_panel_solo(const DenseWs &w) { return w.pctl && gpx_panel_solo_np(w.np, w.batch); }

size_t gpx_panel_ctl_bytes()
{
    return (size_t)(PCTL_GATES + 2) * sizeof(int);
}

int *gpx_panel_gates(const DenseWs &w) { return w.pctl ? w.pctl + PCTL_GATES : nullptr; }

int gpx_panel(hipStream_t s, const DenseWs &w, int off, int n, int extra, int gate_need0,
              int gate_need1, bool rhs)
{
    const int T = n / 128, E = extra / 128;
    // a diagonal block of at most GPX_PANEL_MAX, or (round 3) a whole matrix of at most
    // GPX_PANEL_WHOLE_MAX: every tile of the factorisation as a task of this one launch
    const bool whole = n > GPX_PANEL_MAX;
    // (a whole matrix -- also one of at most GPX_PANEL_MAX that is a single panel, round 4:
    // `rhs` -- may have ONE more tile column in the padding right of it: a right-hand side
    // in its first column, which comes back as R^-T times it -- the forward substitution as
    // tasks of the factorisation)
    const bool aug = (whole || rhs) && extra == 128;
    if (rhs && !aug) {
        gpx_set_error("panel: a right-hand side is one tile column");
        return -1;
    }
    if (n % 128 || T < 2 || n > GPX_PANEL_WHOLE_MAX || (whole && ((extra != 0 && !aug) || off != 0)) ||
        (rhs && off != 0) ||
        !w.pctl || extra % 128 || extra < 0 || extra > GPX_PANEL_MAX ||
        (aug ? (n != w.np || w.ld < n + 128) : off + n + extra > w.np)) {
        gpx_set_error("panel: bad block (order %d, %d more columns)", n, extra);
        return -1;
    }
    GPX_TRY(gpx_test_jitter(s));
    // workgroups beside the spine: the row-panel tasks hold up to seven of them for the
    // length of a leaf and a workgroup that has claimed a task waits for it, whatever else
    // is ready (32 -> 64 -> 128: 450 / 412 / 370 us per 1024-block; N = 4096 evaluation
    // 3.08 -> 2.96 ms with 128); above N = 4096 the chain hides under the products of the
    // same evaluation and every workgroup here holds a whole CU (157 KB of LDS) that the
    // products cannot use while it polls: 32 (round 3, N = 16384: 8 / 16 / 24 / 32 / 48 /
    // 64 workers 72.8 / 71.5 / 70.8 / 71.0 / 71.5 / 71.8 ms per evaluation, N = 8192:
    // 32 / 64 / 128 workers 11.29 / 11.48 / 11.79 ms); on the 32 reserved CUs 29, so
    // that the whole grid is resident whatever order the workgroups are dispatched in
    static const int workers_env = [] {
        const int v = env_once("GPX_PANEL_WG", -1);
        return v < 1 || v > 256 ? -1 : v;
    }();
    static const int timeout_ms = [] {
        const int v = env_once("GPX_PANEL_TIMEOUT_MS", 2000);
        return v < 1 ? 2000 : v;
    }();
    // wide panels (E > 0) carry three times the product tasks and up to 15 row-panel
    // tasks per leaf: 96 / 64 workers (never more than 96: their tasks may wait for
    // another stream's launches, which need CUs of their own)
    static const int wide_env = [] {
        const int v = env_once("GPX_PANEL_WG_WIDE", -1);
        return v < 1 || v > 96 ? -1 : v;
    }();
    // a whole matrix: the trailing updates of all steps are tasks of this launch (21 800 of
    // them at n = 4096, ~9 us each) and have to keep up with a chain of 41 us per tile
    static const int whole_env = [] {
        const int v = env_once("GPX_PANEL_WG_WHOLE", -1);
        return v < 1 || v > 250 ? -1 : v;
    }();
    int workers = whole ? (whole_env > 0 ? whole_env : (T <= 16 ? 160 : 250))
                  : E > 0 ? (wide_env > 0 ? wide_env : (w.np <= 4096 ? 96 : 64))
                  : workers_env > 0 ? workers_env
                  : (w.crit_only && s == w.crit_only) ? 29      // + 3 spine = the 32 CUs
                  : w.np <= 4096 ? 128 : 32;
    // Member-batched launch: the same graph for every member of the workspace. The chain of
    // a member keeps 1-3 spine workgroups busy and its products a handful of workers, so the
    // members share one pool of workers (the interleaved queue) and the launch is sized to
    // the GPU: about 250 workgroups in all -- each holds a whole CU -- of which the spines
    // take 3 per member up to 16 members, 2 up to 40, 1 beyond (the chain of a member then
    // runs solve, diagonal update and leaf one after the other on one CU: 70 instead of 42 us
    // per tile, for a third of the CUs). GPX_PANEL_MSPINE / GPX_PANEL_MWG override.
    const int nmem = w.batch > 1 ? w.batch : 1;
    // (round 5, split spine: per tile a solving and a following task -- five spine workgroups
    // for one matrix, so that solve, follower + leaf and the leaf's inverse tail of
    // neighbouring tiles never wait for each other's workgroup)
    static const int nspine_env = [] {
        const int v = env_once("GPX_PANEL_NSPINE", -1);
        return v < 1 || v > 8 ? -1 : v;
    }();
    static const int stream_env = env_once("GPX_PANEL_STREAM", 1);
    const bool split = panel_split(stream_env != 0, E, aug);
    int nspwg_want = nspine_env > 0 ? nspine_env : (split ? (panel_fold() ? 9 : 5) : 3);
    if (nmem > 1) {
        static const int mspine_env = [] {
            const int v = env_once("GPX_PANEL_MSPINE", -1);
            return v < 1 || v > 9 ? -1 : v;
        }();
        static const int mwg_env = [] {
            const int v = env_once("GPX_PANEL_MWG", -1);
            return v < 8 || v > 1024 ? -1 : v;
        }();
        nspwg_want = mspine_env > 0 ? mspine_env : (nmem <= 16 ? 3 : (nmem <= 40 ? 2 : 1));
        const int total = mwg_env > 0 ? mwg_env : 250;
        workers = std::max(8, total - nmem * nspwg_want);
        if (E > 0 && !aug) {
            gpx_set_error("panel: wide panels are not member-batched");
            return -1;
        }
    }
    // the schedule (a topological order) is simulated for one member's share of the workers
    const int sched_workers = nmem > 1 ? std::min(128, std::max(4, workers / nmem)) : workers;
    PanelList pl;
    // (full_w: a launch over a whole matrix that leaves ALL of R^-1 behind)
    const int ig = w.full_w && off == 0 && n == w.np && T > PANEL_IG ? T : PANEL_IG;
    // one workgroup per member, the member's whole graph on it (gpx_panel_solo)
    const bool solo = nmem > 1 && off == 0 && n == w.np && (E == 0 || aug) && gpx_panel_solo(w);
    GPX_TRY(panel_list(T, E, sched_workers, aug, ig, &pl, solo ? (w.no_inverse ? 2 : 1) : 0));
    if (solo) {
        nspwg_want = 1;
        workers = 0;
    }
    const size_t o = (size_t)off * w.ld + off;
    PanelArgs p;
    p.bA = w.A + o;
    p.bW = w.W + o;
    p.bX = w.Kinv + o;
    p.ld = w.ld;
    p.tasks = pl.dev;
    p.spine = pl.dev + pl.ntasks;
    p.ntasks = pl.ntasks;
    p.nspine = pl.nspine;
    p.nspwg = std::min(nspwg_want, pl.nspine);
    if (!solo) {      // (a solo workgroup waits for nobody: any residency makes progress)
        // Co-residency (the progress argument of this launch): every spine workgroup -- they
        // come first in the grid -- and at least one worker must be resident at the same time,
        // and each holds a whole CU. Reachable only through the environment switches today
        // (GPX_GROUP_MEMBERS up to 256 with the sweep off): fewer spines per member, or an
        // error instead of a launch that would wait out its 2-s bound.
        static int ncu_of[64] = {};
        int device = 0;
        GPX_HIP(hipGetDevice(&device));
        int ncu = 256;
        if (device >= 0 && device < 64) {
            if (!ncu_of[device]) {
                hipDeviceProp_t prop;
                GPX_HIP(hipGetDeviceProperties(&prop, device));
                ncu_of[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            ncu = ncu_of[device];
        }
        while (p.nspwg > 1 && !gpx_panel_grid_fits(nmem, p.nspwg, ncu)) --p.nspwg;
        if (!gpx_panel_grid_fits(nmem, p.nspwg, ncu)) {
            gpx_set_error("panel: %d members need %d spine workgroups, the device has %d CUs "
                          "(run such groups through the lock-step sweep)", nmem, nmem * p.nspwg, ncu);
            return -1;
        }
    }
    p.nctr = pl.nctr;
    p.nmem = nmem;
    p.mstride = nmem > 1 ? w.mstride : 0;
    p.pstride = nmem > 1 ? w.pstride : 0;
    if (nmem > 1 && (w.pstride < PCTL_HEAD + pl.nctr || w.mstride <= 0)) {
        gpx_set_error("panel: bad member strides (%lld, %d)", w.mstride, w.pstride);
        return -1;
    }
    p.gates = w.pctl + PCTL_GATES;
    p.gate_need0 = gate_need0;
    p.gate_need1 = gate_need1;
    p.ctl = w.pctl;
    p.info = w.info;
    p.goff = off;
    // (GPX_PANEL_TIMEOUT_US: test hook -- a bound of a few microseconds makes every launch of
    // two or more tiles end in "timed out waiting", deterministically; tools/check_safe_mode.py)
    static const int timeout_us = env_once("GPX_PANEL_TIMEOUT_US", 0);
    p.timeout = timeout_us > 0 ? (long long)timeout_us * 100LL : (long long)timeout_ms * 100000LL;
    static const int strict = env_once("GPX_PANEL_STRICT", 0);
    p.strict = strict;
    static const int leafskip = env_once("GPX_PANEL_LEAF_SKIP", 0) |
                                (env_once("GPX_LEAF_MFMA", 1) ? 0 : 32);
    // (solo, value-only members: nothing reads W beyond the diagonal 16-blocks of the solves)
    p.leafskip = leafskip | (solo && w.no_inverse ? 8 : 0);
    p.dbg = nullptr;
    p.trace = nullptr;
    static const int debug = env_once("GPX_PANEL_DEBUG", 0);
    static int *dbg_host = nullptr;          // developer runs are single-threaded
    static long long *trace_dev = nullptr;
    // + the spine workgroups
    const int grid = (int)std::min<long long>(workers, (long long)pl.ntasks * nmem) + p.nspwg * nmem;
    if (debug && (nmem == 1 || solo)) {
        if (!dbg_host) GPX_HIP(hipHostMalloc((void **)&dbg_host, 264 * 8 * sizeof(int)));
        memset(dbg_host, 0xff, 264 * 8 * sizeof(int));
        p.dbg = dbg_host;
        if (debug >= 2 && pl.ntasks + pl.nspine <= 32768) {      // (a whole 4096-matrix: 9 600)
            if (!trace_dev) GPX_HIP(hipMalloc((void **)&trace_dev, 32768 * 32 * sizeof(long long)));
            GPX_HIP(hipMemsetAsync(trace_dev, 0, 32768 * 32 * sizeof(long long), s));
            p.trace = trace_dev;
        }
    }
    // One panel launch at a time per device (round 4). Its workgroups hold a whole CU each and
    // spin on each other's counters; a task only waits for tasks that resident workgroups have
    // claimed, so ONE launch always makes progress -- but two launches from two streams can
    // each have their workers resident and a spine workgroup queued behind the other's (the
    // dispatcher deals blocks to the XCDs round robin and fills each XCD on its own), and then
    // neither moves until the 2-s limit aborts them: seen with four host threads driving a
    // handle each (tools/soak_threads.py: "the panel kernel timed out" in two of them).
    // Every launch therefore waits, on the device, for the launch before it on another
    // stream (one event per device; a wait captures the record made before it). Launches of
    // one stream -- an evaluation's look-ahead, a group -- follow each other anyway.
    // Processes are not ordered against each other: one process per GPU.
    // GPX_PANEL_SERIAL=0: no ordering (rounds 1-3).
    if (stream_env != 0 && T >= 2) {
        // the tiles that are polled as data: the mailboxes of the diagonal tiles (diagonal
        // tiles of the staging matrix: unused -- the first K^-1 update that writes them comes
        // after this block's factorisation) and, with the split spine, the tiles (s, s+1) the
        // followers read (dead storage of A here: an unfactored off-diagonal tile lives in the
        // staging area until its row-panel step writes R into A)
        const int TWl = T + E;
        hipLaunchKernelGGL(panel_sentinel_kernel, dim3(TWl * TWl, nmem, 4), dim3(256), 0, s, p.bA,
                           p.bX, T, TWl, split ? 1 : 0, p.ld, p.mstride);
        GPX_HIP(hipGetLastError());
    }
    {
        static const int serial = env_once("GPX_PANEL_SERIAL", 1);
        struct Last { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; };
        static Last last[64];
        static std::mutex mu;
        int device = 0;
        GPX_HIP(hipGetDevice(&device));
        // (solo launches wait for no other workgroup and cannot starve or be starved)
        if (serial && !solo && device >= 0 && device < 64) {
            std::lock_guard<std::mutex> lock(mu);
            Last &l = last[device];
            if (!l.ev) GPX_HIP(hipEventCreateWithFlags(&l.ev, hipEventDisableTiming));
            if (l.stream && l.stream != s) GPX_HIP(hipStreamWaitEvent(s, l.ev, 0));
            hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(256), LEAF2_LDS, s, p);
            GPX_HIP(hipGetLastError());
            GPX_HIP(hipEventRecord(l.ev, s));
            l.stream = s;
        } else {
            hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(256), LEAF2_LDS, s, p);
            GPX_HIP(hipGetLastError());
        }
    }
    if (debug && (nmem == 1 || solo)) {     // developer aid: watch the launch, dump the progress log if it stalls
        for (int ms = 0; ms < 3000; ++ms) {
            if (hipStreamQuery(s) == hipSuccess) {
                if (debug >= 2 && p.trace) {
                    const int nall = pl.ntasks + pl.nspine;
                    std::vector<long long> tr(32 * nall);
                    std::vector<PTask> tk(nall);
                    GPX_HIP(hipMemcpy(tr.data(), trace_dev, tr.size() * 8, hipMemcpyDeviceToHost));
                    GPX_HIP(hipMemcpy(tk.data(), pl.dev, tk.size() * sizeof(PTask),
                                      hipMemcpyDeviceToHost));
                    long long base = tr[0];
                    for (int i = 0; i < nall; ++i) base = std::min(base, tr[32 * i]);
                    fprintf(stderr, "panel trace T=%d tasks=%d (us: claim start end | wg op k sig)\n",
                            T, nall);
                    for (int i = 0; i < nall; ++i)
                        fprintf(stderr, "  %4d %8.2f %8.2f %8.2f | %2lld %d %4d %3d | %.2f %.2f %.2f %.2f\n",
                                i, (tr[32 * i] - base) * 0.01, (tr[32 * i + 1] - base) * 0.01,
                                (tr[32 * i + 2] - base) * 0.01, tr[32 * i + 3], tk[i].op,
                                tk[i].khi - tk[i].klo, (int)tk[i].sig,
                                tr[32 * i + 4] ? (tr[32 * i + 4] - base) * 0.01 : 0.0,
                                tr[32 * i + 5] ? (tr[32 * i + 5] - base) * 0.01 : 0.0,
                                tr[32 * i + 6] ? (tr[32 * i + 6] - base) * 0.01 : 0.0,
                                tr[32 * i + 7] ? (tr[32 * i + 7] - base) * 0.01 : 0.0);
                    {   // in-kernel clock: shader cycles (s_memtime) per 100-MHz tick
                        double cyc = 0.0, ticks = 0.0;
                        for (int i = 0; i < nall; ++i) {
                            cyc += (double)(tr[32 * i + 13] - tr[32 * i + 12]);
                            ticks += (double)(tr[32 * i + 2] - tr[32 * i + 1]);
                        }
                        fprintf(stderr, "  in-kernel clock %.0f MHz\n", cyc / ticks * 100.0);
                    }
                    for (int i = 0; i < nall; ++i)
                        if (tr[32 * i + 16]) {
                            fprintf(stderr, "  leaf %4d", i);
                            for (int q = 16; q < 31; ++q)
                                fprintf(stderr, " %.2f", tr[32 * i + q] ? (tr[32 * i + q] - base) * 0.01 : 0.0);
                            fprintf(stderr, "\n");
                        }
                    for (int i = 0; i < nall; ++i)
                        if (tr[32 * i + 8])
                            fprintf(stderr, "  sub %4d %.2f %.2f %.2f %.2f\n", i,
                                    (tr[32 * i + 8] - base) * 0.01, (tr[32 * i + 9] - base) * 0.01,
                                    (tr[32 * i + 10] - base) * 0.01, (tr[32 * i + 11] - base) * 0.01);
                }
                return 0;
            }
            usleep(1000);
        }
        fprintf(stderr, "panel stalled: T=%d tasks=%d grid=%d\n", T, pl.ntasks, grid);
        for (int g = 0; g < grid; ++g)
            fprintf(stderr, "  wg %2d task %4d state %d dep %d need %d\n", g, dbg_host[8 * g],
                    dbg_host[8 * g + 1], dbg_host[8 * g + 2], dbg_host[8 * g + 3]);
        fflush(stderr);
        _exit(3);
    }
    return 0;
}
