// Device code of the 128x128 diagonal leaf (see leaf.hip for the description),
// shared by the stand-alone leaf kernel and the diagonal-panel kernel (panel.hip).
#pragma once
#include "gpx_internal.h"

#ifndef GPX_V4D
#define GPX_V4D
typedef double v4d __attribute__((ext_vector_type(4)));
#endif

#define LB GPX_TILE
#define NBK 8                          // 16-blocks per side
#define LS 138                         // LDS row stride of the block (doubles): even, and 16 rows
                                       // land on distinct banks for the row-strided fragment reads
#define YS 17
#define LEAF2_LDS ((LB * LS + NBK * 256 + 16 * YS + 128) * 8)  // + pivot-row / panel buffers

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// A(p): factor the diagonal block p in the registers of one wave (lanes 16-63
// mirror lanes 0-15). Writes U into S (zeros below the diagonal), Y = U^-T into
// Ys[i][j] and U^-1 = Y^T into Wd[p][row][col].
// The scaled pivot row goes through a 16-double LDS buffer and comes back as
// broadcast reads. Fetching the 15 multipliers of a step with v_readlane instead
// keeps 30 SGPRs live per step; unrolled over 16 steps that spilled ~460 SGPRs to
// VGPR lanes (writelane / readlane pairs and their hazard nops): 26 of the leaf's
// 55 us.
__device__ __forceinline__ void diag_factor(double *__restrict__ S, int p,
                                            double *__restrict__ Wd,
                                            double *__restrict__ Ys,
                                            double *__restrict__ Rb, int lane,
                                            int *__restrict__ info, int goff, bool &bad)
{
    // lanes 0-15: column jj of the block; lanes 16-31: column jj of the identity, which
    // the same row operations turn into Y = U^-T. A VALU instruction costs the same
    // for 16 or 64 active lanes, so both run in ONE instruction stream (half the FMAs
    // of two register arrays). Lanes 32-63 mirror lanes 0-31.
    const int jj = lane & 15;
    const bool ident = (lane & 16) != 0;
    const int i0 = 16 * p;
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double a = S[(i0 + r) * LS + i0 + jj];
        v[r] = ident ? ((r == jj) ? 1.0 : 0.0) : a;
    }
    // Pivot chain: readlane -> rsqrt -> scale -> update of the next row -> readlane ...
    // Only the next TWO rows get the update of pivot k at once, their multipliers fetched
    // with v_readlane (4 SGPRs); the other rows take it one pivot later, from the scaled
    // pivot row that went through a 16-double LDS buffer in the meantime. A wave issues
    // in order: this keeps the LDS round trip of the multipliers (write, broadcast reads,
    // wait) out of pivot k + 1's chain. Measured: 5050 cycles per 16x16 block (316 per
    // pivot, ~52 instructions each) either way -- the step is bound by instruction issue,
    // not by that round trip.
    // Invariant at the top of step k: rows k and k+1 carry every pivot < k, rows >= k+2
    // every pivot < k-1. (A non-positive pivot turns into NaN and is found after the loop;
    // v_rsq_f64 + one correction step, no special-case selects.)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double piv = readlane_f64(v[k], k);
        // unscaled multipliers of the next two rows, fetched while the rsqrt runs
        const double a1 = readlane_f64(v[k], k < 15 ? k + 1 : k);
        const double a2 = readlane_f64(v[k], k < 14 ? k + 2 : k);
        const double y0 = __builtin_amdgcn_rsq(piv);
        if (k >= 1) {                                     // pivot k-1 on rows >= k+2
            const double *rb = Rb + 32 * ((k - 1) & 1);
            double u[16];
#pragma unroll
            for (int i = k + 2; i < 16; ++i) u[i] = rb[i];            // U[k-1][i], broadcast
#pragma unroll
            for (int i = k + 2; i < 16; ++i) v[i] -= u[i] * v[k - 1];
        }
        const double e = fma(-piv * y0, y0, 1.0);
        const double rinv = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        // lanes left of the diagonal carry garbage from here on (never read by other
        // lanes, zeroed when the block is stored): no selects in the chain
        v[k] *= rinv;
        if (k < 15) {
            // slots 0-15: row k of U; 16-31: row k of Y, unused. (Writing only from the
            // block lanes under an exec mask gave wrong multipliers on gfx950 -- every
            // lane stores.) Two buffers, alternating: no WAR wait
            Rb[32 * (k & 1) + (lane & 31)] = v[k];
            v[k + 1] -= (a1 * rinv) * v[k];
            if (k < 14) v[k + 2] -= (a2 * rinv) * v[k];
        }
    }
    // U[jj][jj] = sqrt(pivot jj); NaN from the first non-positive pivot on
    double dg = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) dg = (r == jj) ? v[r] : dg;
    const unsigned long long fail = __ballot(!(dg > 0.0)) & 0xFFFFull;
    if (fail != 0 && !bad) {
        bad = true;
        if (lane == 0) atomicCAS(info, 0, goff + i0 + __ffsll((long long)fail));
    }
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[(i0 + r) * LS + i0 + jj] = (r <= jj) ? v[r] : 0.0;
    } else if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Ys[r * YS + jj] = v[r];                       // Y[r][jj]
            Wd[p * 256 + jj * 16 + r] = v[r];             // U^-1[jj][r] = Y[r][jj]
        }
    }
}


// A(p) on the matrix pipe (round 3). The register factorisation above spends ~52
// instructions per pivot -- 316 cycles, of which the dependent chain (rsq, correction,
// scale, first row update) is ~100 -- because every pivot applies its rank-1 update to
// all later rows with one FMA per row. Here the block is blocked by 4: the 16 x 16 block
// and the identity beside it live in the accumulator layout of v_mfma_f64_16x16x4 (lane
// (lr, lk), register r: row lk + 4r, column lr); for each panel of four rows the rows
// are gathered into every lane group through a 1-KB LDS buffer (column per lane, as
// above, replicated four times so that the pivots need no cross-group traffic), factored
// there -- only the at most three later rows OF THE PANEL take each rank-1 update -- and
// the rank-4 update of everything below the panel is ONE MFMA for the block and one for
// the identity side: A[i][kk] = U[4q + kk][i] is the gathered panel itself, picked by lane
// group. Entries left of the diagonal carry garbage as before; it only ever reaches rows
// and columns that are final or never read.
// `from` >= 0: the last update of the block, -= R[from][p]^T R[from][p] (row panel at rows
// `from` of S), is applied here, straight into the accumulator layout the factorisation
// works in, instead of an LDS store and reload in front of it.
// (Measured and dropped: taking the in-panel updates from the UNSCALED pivot row with the
// multiplier m / pivot -- v_rcp_f64 + correction, so that the reciprocal square root and
// the scaling leave the dependent chain, 7 operations per pivot instead of 9 -- was slower,
// 1.88 against 1.72 us per 16 x 16 block: a lone wave pays for every instruction it
// issues, and the second transcendental and its correction cost more than the two
// chain links they remove.)
__device__ __forceinline__ void diag_factor_mfma(double *__restrict__ S, int p,
                                                 double *__restrict__ Wd,
                                                 double *__restrict__ Ys,
                                                 double *__restrict__ Pb, int lane,
                                                 int *__restrict__ info, int goff, bool &bad,
                                                 int from = -1)
{
    const int lr = lane & 15, lk = lane >> 4;
    const int i0 = 16 * p;
    v4d a, y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = S[(i0 + lk + 4 * r) * LS + i0 + lr];
        y[r] = (lk + 4 * r == lr) ? 1.0 : 0.0;
    }
    if (from >= 0) {
        double o[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) o[ks] = S[(from + 4 * ks + lk) * LS + i0 + lr];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(-o[ks], o[ks], a, 0, 0, 0);
    }
    double dg = 0.0;                                     // U[lr][lr]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // rows 4q + t of block and identity side: register q of lane group t -> all groups
        Pb[lane] = a[q];
        Pb[64 + lane] = y[q];
        double pu[4], py[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pu[t] = Pb[16 * t + lr];
            py[t] = Pb[64 + 16 * t + lr];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double piv = readlane_f64(pu[t], 4 * q + t);
            const double y0 = __builtin_amdgcn_rsq(piv);
            const double e = fma(-piv * y0, y0, 1.0);
            const double rinv = fma(y0 * e, fma(e, 0.375, 0.5), y0);
            pu[t] *= rinv;
            py[t] *= rinv;
#pragma unroll
            for (int u = t + 1; u < 4; ++u) {
                const double m = readlane_f64(pu[t], 4 * q + u);     // U[4q+t][4q+u]
                pu[u] -= m * pu[t];
                py[u] -= m * py[t];
            }
        }
        // this lane group's row of the panel: U[4q + lk][lr], Y[4q + lk][lr]
        // (selects kept opaque: written as a plain chain the compiler turns them into an
        // indexed load from a scratch copy of pu[] / py[], a memory round trip on the chain)
        double ua = pu[0], yb = py[0];
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            ua = lk == t ? pu[t] : ua;
            yb = lk == t ? py[t] : yb;
            asm volatile("" : "+v"(ua), "+v"(yb));
        }
        const int row = 4 * q + lk;
        if (row == lr) dg = ua;
        S[(i0 + row) * LS + i0 + lr] = (lr >= row) ? ua : 0.0;
        Ys[row * YS + lr] = yb;                             // Y[row][lr]
        Wd[p * 256 + lr * 16 + row] = yb;                   // U^-1[lr][row] = Y[row][lr]
        if (q < 3) {
            // rows below the panel: block -= U_p^T U_p, identity side -= U_p^T Y_p
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(-ua, ua, a, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(-ua, yb, y, 0, 0, 0);
        }
    }
    // U[jj][jj] = sqrt(pivot jj) sits in lane (jj, jj % 4); NaN from the first non-positive
    // pivot on
    const unsigned long long f64 = __ballot((lr & 3) == lk && !(dg > 0.0));
    const unsigned long long fail = (f64 | (f64 >> 16) | (f64 >> 32) | (f64 >> 48)) & 0xFFFFull;
    if (fail != 0 && !bad) {
        bad = true;
        if (lane == 0) atomicCAS(info, 0, goff + i0 + __ffsll((long long)fail));
    }
}

// B(p): X = Y * B for the 16x16 block at rows i0, columns c0 (in place)
__device__ __forceinline__ void solve_block(double *__restrict__ S,
                                            const double *__restrict__ Ys, int i0, int c0,
                                            int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    double a[4], b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a[ks] = Ys[lr * YS + 4 * ks + lk];                          // Y[i][k]
        b[ks] = S[(i0 + 4 * ks + lk) * LS + c0 + lr];               // B[k][j]
    }
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(i0 + lk + 4 * r) * LS + c0 + lr] = acc[r];
}

// C(p): block (q, r) of the trailing matrix -= R[p][q]^T R[p][r]
__device__ __forceinline__ void update_block(double *__restrict__ S, int i0, int q, int r,
                                             int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    double a[4], b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a[ks] = -S[(i0 + 4 * ks + lk) * LS + 16 * q + lr];          // -R[p][q][k][i]
        b[ks] = S[(i0 + 4 * ks + lk) * LS + 16 * r + lr];           //  R[p][r][k][j]
    }
    v4d acc;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = S[(16 * q + lk + 4 * t) * LS + 16 * r + lr];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) S[(16 * q + lk + 4 * t) * LS + 16 * r + lr] = acc[t];
}

// C(p) for G blocks at once (round 3): the first trailing updates of a leaf (27, 20 and 14
// blocks on three waves) outlasted the MFMA-blocked factorisation they run beside. One
// block at a time is an LDS round trip, four dependent MFMAs and a store (630 cycles for
// 256 of MFMA), two at a time (round 2) still a round trip and a four-deep dependent chain
// per pair; with G chains in flight the matrix pipe issues back to back.
template <int G>
__device__ __forceinline__ void update_blocks(double *__restrict__ S, int i0,
                                              const int (&bq)[5], const int (&br)[5], int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    double a[G][4], b[G][4];
    v4d c[G];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double *row = S + (i0 + 4 * ks + lk) * LS + lr;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            a[g][ks] = -row[16 * bq[g]];
            b[g][ks] = row[16 * br[g]];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) c[g][t] = S[(16 * bq[g] + lk + 4 * t) * LS + 16 * br[g] + lr];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int g = 0; g < G; ++g)
            c[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][ks], b[g][ks], c[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) S[(16 * bq[g] + lk + 4 * t) * LS + 16 * br[g] + lr] = c[g][t];
}

// 16x16 block (I, J) of the upper-triangular inverse being assembled: diagonal
// blocks come from Wd, the others from S (where W12 blocks overwrite R12 blocks)
__device__ __forceinline__ const double *wblock(const double *S, const double *Wd, int I,
                                                int J, int &stride)
{
    if (I == J) {
        stride = 16;
        return Wd + I * 256;
    }
    stride = LS;
    return S + (16 * I) * LS + 16 * J;
}

// one doubling level of the inverse for this wave's column strip: node t, m
// blocks per half, strip c -> block column J
template <int M>
__device__ __forceinline__ void inverse_level(double *__restrict__ S,
                                              const double *__restrict__ Wd, int wave,
                                              int lane)
{
    const int lr = lane & 15, lk = lane >> 4;
    const int t = wave / M, c = wave % M;
    const int base = 2 * M * t;               // first block of the node
    const int J = base + M + c;               // this strip's block column
    v4d T[M];
    // T[kb] = sum_{mb <= c} R[base+kb][base+M+mb] * W22[mb][c]
#pragma unroll
    for (int kb = 0; kb < M; ++kb) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int mb = 0; mb <= c; ++mb) {
            int ws;
            const double *wb = wblock(S, Wd, base + M + mb, J, ws);
            const double *rb = S + (16 * (base + kb)) * LS + 16 * (base + M + mb);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = rb[lr * LS + 4 * ks + lk];            // R[i][k]
                const double b = wb[(4 * ks + lk) * ws + lr];          // W22[k][j]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
        T[kb] = acc;
    }
    __syncthreads();                          // every strip has read R12
    // W12[ib] = -sum_{kb >= ib} W11[ib][kb] * T[kb]; T is already a B operand
#pragma unroll
    for (int ib = 0; ib < M; ++ib) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < M; ++kb) {
            if (kb < ib) continue;
            int ws;
            const double *wb = wblock(S, Wd, base + ib, base + kb, ws);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = -wb[lr * ws + 4 * ks + lk];           // -W11[i][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[kb][ks], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            S[(16 * (base + ib) + lk + 4 * r) * LS + 16 * J + lr] = acc[r];
    }
    __syncthreads();
}

// 16-B global accesses of the leaf. AGENT: agent-scope (sc1) accesses that are
// coherent between workgroups of a running kernel without cache write-back /
// invalidate fences -- what the panel kernel's tile hand-offs need. They go out as
// buffer_load / buffer_store_dwordx4 ... sc1: a 16-B sc1 access runs at the plain rate,
// the 8-B form that __hip_atomic_load / _store lower to at 0.54-0.70x (loads) and 2.7x
// the time per byte (stores), and a lone workgroup moving 128-KB tiles is bound by
// exactly that (MI355X_MICROARCH.md, inter-workgroup visibility table).
typedef int gpx_v4i __attribute__((ext_vector_type(4)));

// raw buffer descriptor over 2 GB starting at p (p wave-uniform); offsets in bytes
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const double *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double2 agent_load16(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    const gpx_v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */);
    return __builtin_bit_cast(double2, v);
}
__device__ __forceinline__ void agent_store16(__amdgpu_buffer_rsrc_t r, int byte_off, double2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gpx_v4i, v), r, byte_off, 0,
                                           16 /* sc1 */);
}

// one wave has stored a panel: drain its stores, then move the counter
__device__ __forceinline__ void leaf_stream_signal(int *stream, int lane, int strict)
{
    __builtin_amdgcn_s_waitcnt(0);
    if (strict) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0)
        __hip_atomic_fetch_add(stream, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the whole leaf for one 256-thread workgroup; smem_raw: LEAF2_LDS bytes of LDS.
// preloaded: skip the block-in phase. stream (panel kernel, AGENT only): counter that
// follows the factorisation. Row panel p
// of R (16 rows, final) and the inverse of its diagonal 16-block go out as soon as they
// exist and the counter then moves to p + 1, so that the workgroups solving the tiles
// to the right of this one (xs_run, panel.hip) work alongside the pivot chain instead
// of after it; strict: release fence before the counter moves.
template <bool AGENT = false>
__device__ __forceinline__ void leaf2_run(double *__restrict__ A, int lda,
                                          double *__restrict__ W, int ldw,
                                          int *__restrict__ info, int goff, int skip,
                                          char *smem_raw, int *stream = nullptr,
                                          int strict = 0, bool preloaded = false,
                                          long long *tr = nullptr, double *M = nullptr)
{
    // skip: timing experiments only (bit 0 diagonal factor, 1 panel solve,
    // 2 trailing update, 3 inverse, 4 streamed stores); 0 in production
    double *S = reinterpret_cast<double *>(smem_raw);       // [LB][LS]
    double *Wd = S + LB * LS;                               // [NBK][16][16]
    double *Ys = Wd + NBK * 256;                            // [16][YS]
    double *Rb = Ys + 16 * YS;                              // [2][32] pivot rows / [2][64] panel

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // M (panel kernel, round 5): the tile's MAILBOX, a scratch tile with the row stride of A
    // that holds a pattern no computation produces until the leaf's stores land. Every
    // streamed panel -- the blocks right of the diagonal 16-block and that block's inverse, at
    // the diagonal position -- goes there as well, and the solves to the right poll THAT
    // data instead of the stream counter (xs_run): a panel is seen one store-to-load latency
    // after it left, not a drain, an atomic and a counter poll later.
    __amdgpu_buffer_rsrc_t rA = agent_rsrc(A), rW = agent_rsrc(W), rM = agent_rsrc(M ? M : A);
    auto gload = [&](int row, int col) -> double2 {
        if (AGENT) return agent_load16(rA, (row * lda + col) * 8);
        return *reinterpret_cast<const double2 *>(A + (size_t)row * lda + col);
    };
    auto gstore = [&](bool toW, int row, int col, double2 v) {
        if (AGENT) {
            agent_store16(toW ? rW : rA, (row * (toW ? ldw : lda) + col) * 8, v);
        } else {
            double *q = toW ? W + (size_t)row * ldw + col : A + (size_t)row * lda + col;
            *reinterpret_cast<double2 *>(q) = v;
        }
    };

    // block in: 32 x 16-B loads per thread, all in flight at once (a plain element
    // loop is one global round trip per iteration for a lone workgroup; two batches
    // of 16 cost 6 us more per leaf; reading only the upper 16-blocks changes nothing:
    // the phase is latency-, not byte-bound)
    // (preloaded: the block is in S already, left there by the task that applied its last
    // update -- xs_run, panel.hip)
    if (!preloaded) {
        double2 tmp[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e2 = tid + 256 * i;
            tmp[i] = gload(e2 >> 6, 2 * (e2 & 63));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e2 = tid + 256 * i;
            *reinterpret_cast<double2 *>(S + (e2 >> 6) * LS + 2 * (e2 & 63)) = tmp[i];
        }
    }
    __syncthreads();

    bool bad = false;
    // skip & 32: the register factorisation of rounds 1-2 instead of the MFMA-blocked one
    if (wave == 0 && !(skip & 1)) {
        if (skip & 32) diag_factor(S, 0, Wd, Ys, Rb, lane, info, goff, bad);
        else diag_factor_mfma(S, 0, Wd, Ys, Rb, lane, info, goff, bad);
    }
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NBK; ++p) {
        const int i0 = 16 * p;
        if (!(skip & 2))
            for (int q = p + 1 + wave; q < NBK; q += 4) solve_block(S, Ys, i0, 16 * q, lane);
        __syncthreads();
        if (tr && tid == 0) tr[16 + 2 * p] = wall_clock64();      // solves of panel p done
        if (AGENT && stream && wave >= 1 && !(skip & 16)) {
            // Row panel p is final. What the solves to the right need goes out now: the
            // blocks right of the diagonal and the inverse of the diagonal block, a third
            // of the columns per wave (waves 1-3; wave 0 is the pivot chain). Each wave
            // moves the counter by one a step later, when its stores have long been
            // acknowledged: panel p is published when the counter reads 3 (p + 1).
            if (p > 0) leaf_stream_signal(stream, lane, strict);
            const int nc2 = 8 * (NBK - 1 - p);            // 16-B chunks per row right of the block
            const int r = i0 + (lane >> 2);
            for (int cc = (lane & 3) + 4 * (wave - 1); cc < nc2; cc += 12) {
                const int c = i0 + 16 + 2 * cc;
                const double2 v = *reinterpret_cast<const double2 *>(S + r * LS + c);
                if (M) agent_store16(rM, (r * lda + c) * 8, v);     // first: the solves wait for it
                gstore(false, r, c, v);
            }
            if (wave == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = lane + 64 * i, rr = e >> 3, c = 2 * (e & 7);
                    const double2 v = *reinterpret_cast<const double2 *>(Wd + p * 256 + rr * 16 + c);
                    if (M) agent_store16(rM, ((i0 + rr) * lda + i0 + c) * 8, v);
                    gstore(true, i0 + rr, i0 + c, v);
                }
            }
        }
        if (p == NBK - 1) break;
        if (wave == 0) {
            if (skip & 32) {
                if (!(skip & 4)) update_block(S, i0, p + 1, p + 1, lane);
                if (!(skip & 1)) diag_factor(S, p + 1, Wd, Ys, Rb, lane, info, goff, bad);
            } else if (!(skip & 1)) {
                diag_factor_mfma(S, p + 1, Wd, Ys, Rb, lane, info, goff, bad,
                                 (skip & 4) ? -1 : i0);
            }
        } else if (!(skip & 4)) {
            // blocks (q, r), q <= r, of the trailing matrix except the next diagonal one, in
            // row-major order, every third one to each of waves 1-3, in groups of at most
            // five, evenly sized (nine blocks: 5 + 4, seven: 4 + 3). (Walking the order with
            // a stride of three; the round-2 search loop over all blocks with idx % 3 was
            // ~70 scalar cycles per candidate, 0.8 us of the first update.)
            const int nb = NBK - 1 - p, total = nb * (nb + 1) / 2 - 1;
            const int cnt = (total + 3 - wave) / 3, ngroups = (cnt + 4) / 5;
            const int gs = ngroups ? (cnt + ngroups - 1) / ngroups : 0;
            int q = p + 1, r = p + 1 + wave;
            for (int left = cnt; left > 0;) {
                int bq[5], br[5];
                const int n = min(gs, left);
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (g < n) {
                        while (r >= NBK) {               // past the end of block row q
                            ++q;
                            r += q - NBK;
                        }
                        bq[g] = q;
                        br[g] = r;
                        r += 3;
                    } else {
                        bq[g] = br[g] = NBK - 1;
                    }
                }
                if (n == 5) update_blocks<5>(S, i0, bq, br, lane);
                else if (n == 4) update_blocks<4>(S, i0, bq, br, lane);
                else if (n == 3) update_blocks<3>(S, i0, bq, br, lane);
                else if (n == 2) update_blocks<2>(S, i0, bq, br, lane);
                else update_blocks<1>(S, i0, bq, br, lane);
                left -= n;
            }
        }
        __syncthreads();
        if (tr && tid == 0) tr[17 + 2 * p] = wall_clock64();      // trailing update p done
    }
    if (AGENT && stream && wave >= 1) leaf_stream_signal(stream, lane, strict);
    if (tr && tid == 0) tr[6] = wall_clock64();          // pivot chain done

    // R out: upper 16-blocks from S (diagonal blocks carry their own zeros); when
    // streaming, the blocks right of the diagonal have gone out panel by panel
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const int e2 = tid + 256 * i;
        const int r = e2 >> 6, c = 2 * (e2 & 63);
        double2 v = *reinterpret_cast<const double2 *>(S + r * LS + c);
        if ((c >> 4) < (r >> 4)) v = make_double2(0.0, 0.0);
        if (!(AGENT && stream) || (c >> 4) <= (r >> 4)) gstore(false, r, c, v);
    }
    __syncthreads();

    if (tr && tid == 0) tr[7] = wall_clock64();          // R out issued
    if (!(skip & 8)) {
        inverse_level<1>(S, Wd, wave, lane);
        inverse_level<2>(S, Wd, wave, lane);
        inverse_level<4>(S, Wd, wave, lane);
    }

    if (skip & 8) {
        // (nothing will read W beyond the inverses of the diagonal 16-blocks -- value-only
        // members of a lock-step sweep: 16 KB out instead of 128, a lone CU stores ~23 GB/s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;                    // 8 blocks x 16 rows x 8 double2
            const int b = e >> 7, rr = (e >> 3) & 15, c = 2 * (e & 7);
            gstore(true, 16 * b + rr, 16 * b + c,
                   *reinterpret_cast<const double2 *>(Wd + b * 256 + rr * 16 + c));
        }
        return;
    }
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const int e2 = tid + 256 * i;
        const int r = e2 >> 6, c = 2 * (e2 & 63);
        const int br = r >> 4, bc = c >> 4;
        double2 v = make_double2(0.0, 0.0);
        if (bc > br) v = *reinterpret_cast<const double2 *>(S + r * LS + c);
        else if (bc == br)
            v = *reinterpret_cast<const double2 *>(Wd + br * 256 + (r & 15) * 16 + (c & 15));
        gstore(true, r, c, v);
    }
}

