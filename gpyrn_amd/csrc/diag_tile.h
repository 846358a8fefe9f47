// potrf + inverse of one 128 x 128 diagonal tile by one workgroup (device code; included by factor.hip, smalln.hip and
// the probes under profiles/probes/).  Replaces jnp.linalg.cholesky on a tile (/root/reference/gpyrn/meanfield.py:71-89) and the
// triangular solve against it.
#pragma once
#include "gprn_internal.h"
#include "tile_mma.h"

#define PP 18             // LDS pitch (doubles) of a 16-wide column panel: conflict-free operand fetch
#define NSB 8             // 16x16 sub-blocks per tile edge

// ------------------------------------------------------------------ diag
// potrf + inverse of one 128x128 diagonal tile by one workgroup, blocked by 16 so that
// all O(n^3) work runs on v_mfma_f64_16x16x4_f64 -- and with the whole tile resident in
// the MFMA accumulators: 3 compute waves own the 8 sub-tile rows ({0,7}, {1,6,3},
// {2,5,4}: equal update counts; up to 24 sub-tiles of 16x16 = 192 VGPRs per lane), the
// 4th wave runs the scalar pivot chains (one wave per SIMD: 512 VGPRs each).  Only the
// current 16-wide column panel passes through LDS (2 x 18 KiB), so the kernel fits on
// a CU next to the bulk-update workgroups of the look-ahead stream.
//
// Storage convention (as for the big tiles): sub-tile (P,Q), P >= Q holds B then L;
// P < Q holds the transposed running right-hand side of L X = I, S(P,Q) = R(Q,P)^T.
// With it every step kb is the same formula on sub-tiles:
//   base   (wave 3)   : S(kb,kb) -> L_kb,  X_kb = L_kb^-1 -> XD     (register/shuffle potf2)
//   panel             : S(P,kb) <- S(P,kb) X_kb^T                    every P != kb
//   update            : S(P,Q) -= S(P,kb) S(Q,kb)^T                  Q > kb, P < kb or P >= Q
//                       S(kb,Q)  = -X_kb^T S(Q,kb)^T                 first touch of R's row kb
// The update of column kb+1 goes first (U1) and is published to LDS, so that the base
// wave factors S(kb+1,kb+1) while the compute waves finish the rest of the update (U2).

// 1/sqrt(x) to fp64 round-off from the hardware seed (v_rsq_f64) plus two Newton steps: a
// fraction of the latency of the IEEE sqrt + divide sequences, and this sits on the serial
// pivot chain.  NaN for x < 0 (jnp.linalg.cholesky semantics).
__device__ __forceinline__ double rsqrt_nr(double x)
{
#ifdef GPRN_EXACT_PIVOT           // (accuracy experiment: IEEE sqrt and division)
    return __ddiv_rn(1.0, __dsqrt_rn(x));
#endif
    double y = __builtin_amdgcn_rsq(x);
    const double hx = -0.5 * x;
    y = y * fma(hx * y, y, 1.5);
    y = y * fma(hx * y, y, 1.5);      // second step: seed accuracy is not documented for gfx950 (and dropping
                                      // it does not shorten base16: 7537 vs 7701 cycles, profiles/probes/base16_bench.hip)
    return y;
}

// Lanes of ONE wave talking through LDS: the hardware keeps a wave's LDS operations in order, so no wait is needed
// -- but the compiler must be told that other lanes' stores are visible to this lane's later loads.  A bare
// __builtin_amdgcn_wave_barrier() is not a memory fence for the optimiser: round 2 caught GVN reusing a lane's
// PREVIOUS load of an LDS word that only other lanes had rewritten (loads moved under the writers' exec mask).
// Wavefront-scope fences cost no instruction.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane /* wave-uniform */)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// potf2 + trtri2 of a 16x16 block, the block in ONE MFMA accumulator (C layout: lane (fr = l&15, fk = l>>4), register
// t holds S[fk + 4t][fr]; lower = B then L, strict upper = transposed right-hand side as in S), four pivots per round:
//   1. the round's four columns go to LDS, [row][n] (the block's own rows as identity rows);
//   2. the 4x4 diagonal block comes to every lane with v_readlane (10 values) and is factored by all lanes at once
//      (uniform data), L and the reciprocal pivots; X = L^-1 lane-parallel: lane (., fk) runs the
//      substitution for row fk;
//   3. W = Sp X^T, lane (fr = r, fk = m) forms W[r][m] from its row of the LDS columns and row m of X -- L's panel
//      rows below the block, the inverse's rows k0..k0+3 above it and inside (identity rows: W[k0+i][m] = X[m][i])
//      -- which is the MFMA OPERAND layout;
//   4. the rank-4 update of the whole block is ONE more MFMA, C -= W W^T (A = -W, B = W).  It also touches the
//      not-yet-started part of the right-hand side (k0+4 <= r < b, zero so far); those entries are first used by
//      the round of row r's own block, which clears its accumulator register (t = that round) before its update.
// W is final: it goes straight to St (L) resp. xd / Xg (X); the block's own L entries are stored by lane 0.
// The single wave that runs this is issue-bound, not latency-bound: the lane-owned form below (base16_lanes, rounds
// 1-2) needs ~1500 instructions per block, 3.4 us; this one ~600.
#ifdef BASE16_STAMPS     // profiles/probes/base16_bench.hip: shader-clock stamps inside one call, after `dep` is available
__device__ long long b16_stamps[4][8];
#define B16_STAMP(R, i, dep) do { asm volatile("" :: "v"(dep)); b16_stamps[R][i] = clock64(); } while (0)
#else
#define B16_STAMP(R, i, dep) do {} while (0)
#endif
// c: the block in C layout (what lies above the diagonal is ignored); L -> St (lower), X -> xd and Xg
// ACC (the set-up's factorisation of a PRIOR matrix, cond(K) ~ 1e8): W by substitution against the 4 x 4 block instead of
// the product with its explicit inverse -- see subst16_row below
template <bool ACC = false>
__device__ __forceinline__ void base16_regs(v4d c, double* __restrict__ St /* pitch PP */,
                                            double* __restrict__ xd, gptr_t Xg, int ld,
                                            int* info, int slot, int pivot0,
                                            double* __restrict__ line /* 64 doubles of LDS */)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = (fr <= fk + 4 * t) ? c[t] : 0.0;
    int bad_at = 0;
#pragma unroll
    for (int R = 0; R < 4; ++R) {
        const int k0 = 4 * R;
        B16_STAMP(R, 0, c[R]);
        // ---- 1. columns k0..k0+3 -> line[row * 4 + n]
        if (fr >= k0 && fr < k0 + 4) {
            const int n = fr - k0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                line[(fk + 4 * t) * 4 + n] = (t == R) ? (fk == n ? 1.0 : 0.0) : c[t];
        }
        wave_lds_sync();                        // same wave, in-order LDS: the read sees step 1's writes
        const double2 sp01 = *(const double2*)(line + fr * 4), sp23 = *(const double2*)(line + fr * 4 + 2);
        const double sp[4] = {sp01.x, sp01.y, sp23.x, sp23.y};      // (in flight during step 2)
        // ---- 2. the 4x4 diagonal block, lower part, to every lane; L, 1/pivots
        double d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m <= i; ++m) d[i][m] = readlane_f64(c[R], k0 + m + 16 * i);
        B16_STAMP(R, 1, d[3][3]);
        double inv[4], L[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double t = d[i][i];
#pragma unroll
            for (int m = 0; m < i; ++m) t = fma(-L[i][m], L[i][m], t);
            bad_at = (bad_at == 0 && !(t > 0.0)) ? k0 + i + 1 : bad_at;
            inv[i] = rsqrt_nr(t);
            L[i][i] = t * inv[i];
#pragma unroll
            for (int n = i + 1; n < 4; ++n) {
                double u = d[n][i];
#pragma unroll
                for (int m = 0; m < i; ++m) u = fma(-L[n][m], L[i][m], u);
                L[n][i] = u * inv[i];
            }
        }
        B16_STAMP(R, 2, L[3][3]);
        // row fk of X = L^-1 in every lane (X L = I from the diagonal backwards; zero beyond the diagonal), and
        // with it W[r = fr][m = fk] = sum_n Sp[r][n] X[m][n]
        double xr[4], w = 0.0;
        if constexpr (ACC) {
            // row r of W solves  w L^T = Sp[r][.]  (every lane has the whole row and the uniform L); lane (r, m) keeps w[m]
            double ws[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                double u = sp[n];
#pragma unroll
                for (int k = 0; k < n; ++k) u = fma(-ws[k], L[n][k], u);
                ws[n] = u * inv[n];
            }
            w = fk == 0 ? ws[0] : (fk == 1 ? ws[1] : (fk == 2 ? ws[2] : ws[3]));
        } else {
#pragma unroll
        for (int n = 3; n >= 0; --n) {
            double u = 0.0;
#pragma unroll
            for (int k = n + 1; k < 4; ++k) u = fma(xr[k], L[k][n], u);
            xr[n] = (fk == n) ? inv[n] : -u * inv[n];
            w = fma(sp[n], xr[n], w);
        }
        }
        B16_STAMP(R, 5, w);
        // ---- 4. rank-4 update of what is still to come
        if (R < 3) {
            c[R] = 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-w, w, c, 0, 0, 0);
        }
        B16_STAMP(R, 6, c[3]);
        // ---- results of the round
        const bool below = fr > k0 + 3;
        if (below) St[fr * PP + k0 + fk] = w;
        const double xv = below ? 0.0 : w;
        xd[(k0 + fk) * PP + fr] = xv;
        Xg[(size_t)(k0 + fk) * ld + fr] = xv;
        if (l == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m <= i; ++m) St[(k0 + i) * PP + k0 + m] = L[i][m];
        }
        wave_lds_sync();
        B16_STAMP(R, 7, xv);
    }
    if (bad_at && l == 0 && info[slot] == 0) info[slot] = pivot0 + bad_at;
}

// the block from LDS (St, lower part)
template <bool ACC = false>
__device__ __forceinline__ void base16(double* __restrict__ St /* pitch PP */,
                                       double* __restrict__ xd, gptr_t Xg, int ld,
                                       int* info, int slot, int pivot0,
                                       double* __restrict__ line /* 64 doubles of LDS */)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    v4d c;
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = St[(fk + 4 * t) * PP + fr];
    base16_regs<ACC>(c, St, xd, Xg, ld, info, slot, pivot0, line);
}

// 16x16 tile in MFMA C/D layout <-> LDS image [row][col], pitch PP
__device__ __forceinline__ void put16(double* __restrict__ T, const v4d& v)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) T[(fk + 4 * t) * PP + fr] = v[t];
}
__device__ __forceinline__ v4d get16(const double* __restrict__ T)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    v4d v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = T[(fk + 4 * t) * PP + fr];
    return v;
}

// Schedule.  Phase kb = 0..7, one workgroup barrier in the middle (M) and one at the end (E):
//   compute waves:  panel(kb) with X_kb            | M |  column kb -> global memory (it is final);
//                   (diag sub-tile kb <- L_kb)     |   |  update(kb); publish column kb+1 (pre-scaling) and
//                                                  |   |  diagonal sub-tile kb+2
//   pivot wave   :  L' = S(kb+1,kb) X_kb^T,        | M |  base(kb+1): L_{kb+1}, X_{kb+1}
//                   T = S(kb+1,kb+1) - L' L'^T     |   |
// The pivot wave runs one step ahead of the compute waves: it needs only the column panel and
// the diagonal sub-tile as they stood after update(kb-1), both published to LDS in phase kb-1,
// so the 16-pivot chains (the serial part) never wait for the bulk of the update.
//
// Each compute wave is its own instantiation (its sub-tile rows are compile-time constants) and the phase loop is
// unrolled: a phase is straight-line code, the operands of ALL its products (one 16 x 16 row block of the scaled
// column per sub-tile row, two ds_read_b128 each) are fetched once, and the MFMAs of different sub-tiles alternate
// -- round 2 measured the branchy form (one basic block per 16x16 product, operands re-read for each) at 600 clocks
// per product where the four dependent MFMAs need 256.  Column kb of the result is stored during phase kb instead of
// in an epilogue of its own (5 us of 38 for one tile).
#ifdef DIAG_STAMPS        // profiles/probes/diag_bench.hip: shader-clock stamps per wave, phase and point
__device__ long long diag_stamps[4][NSB + 1][6];
#define DG_STAMP(kb, i) do { if ((threadIdx.x & 63) == 0) diag_stamps[threadIdx.x >> 6][kb][i] = clock64(); } while (0)
#else
#define DG_STAMP(kb, i) do {} while (0)
#endif
// 46.6 KB: the kernel fits on a CU beside two bulk-update workgroups (2 x (41 + 15) KB of the CU's 160) or one with the
// small-batch pad.  (Round 1-2 form: 67 KB with the published column double-buffered -- it is read before barrier M and
// rewritten after it, one buffer does -- and a scratch tile the pivot wave no longer needs.)
#define DIAG_LDS_DOUBLES (128 * PP + 128 * PP + 2 * 16 * PP + 2 * 16 * PP + 64)
// ... and of the ACC form (the pivot wave's own copy of its look-ahead panel block): the set-up's launches only
#define DIAG_LDS_DOUBLES_ACC (DIAG_LDS_DOUBLES + 16 * PP)

struct DiagLds {
    double *PA, *PB, *DG, *XD, *LINE, *LP;
    __device__ explicit DiagLds(double* lds)
        : PA(lds),                       // published column panel (before its scaling)        128 x PP
          PB(PA + 128 * PP),             // current column after scaling by X_kb^T              128 x PP
          DG(PB + 128 * PP),             // diagonal sub-tiles for / from the pivot wave (by parity)
          XD(DG + 2 * 16 * PP),          // X_kb by parity
          LINE(XD + 2 * 16 * PP),
          LP(LINE + 64) {}               // ACC only (DIAG_LDS_DOUBLES_ACC): S(kb+1,kb) L_kb^-T of the pivot wave    16 x PP
};

// ---- ACC: the panel step by SUBSTITUTION.
// The panel step of the blocked factorisation, L_ik = B_ik L_kk^-T, runs everywhere else as a PRODUCT with the explicit
// inverse X_kk (the inverse is what the sweep has to build anyway, and a product is one MFMA pass where a substitution is
// 16 dependent steps).  Its backward error is eps cond(L_kk) |B_ik| where a triangular solve has eps |L_ik| |L_kk^T|: on the
// well-conditioned B = I + D^1/2 K D^1/2 of a sweep (cond 1e3-1e4) that is harmless; on a PRIOR matrix (cond(K) ~ 1e8, a
// pure Periodic kernel: rank-deficient but for the reference's 1e-6 nugget) with a mean far outside the range of K it put
// m^T K^-1 m at 4e-16 cond(K) = 2.8e-8 where LAPACK gets 1.4e-10 (profiles/r05_prior_term_accuracy.txt; the only input of
// the round-5 tests that missed north_star's 1e-8).  The set-up's factorisation of K (meanfield.py:71-89, 621-622; its
// cho_solve quadratic forms :1032, :1050) therefore solves, at all three block levels -- 4 (base16_regs), 16 (here) and 128
// (trsm_rows16, gemm_tile.hip's k_tile_panel<true>) -- as LAPACK's potrf does: row r of the panel solves x L^T = s.
// One lane per row; L (lower, [row][PP]) and the reciprocal pivots (the diagonal of X_kb) are uniform LDS reads; two
// accumulators halve the dependent chain.  S and out may be the same row.
__device__ __forceinline__ void subst16_row(const double* S, const double* __restrict__ Lb, const double* __restrict__ rinv,
                                            int rinv_stride, double* out)
{
    double x[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double2 v = *(const double2*)(S + 2 * j);
        x[2 * j] = v.x; x[2 * j + 1] = v.y;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double a0 = x[c], a1 = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < c; m += 2) {
            a0 = fma(-x[m], Lb[c * PP + m], a0);
            a1 = fma(-x[m + 1], Lb[c * PP + m + 1], a1);
        }
        if (c & 1) a0 = fma(-x[c - 1], Lb[c * PP + c - 1], a0);
        x[c] = (a0 + a1) * rinv[c * rinv_stride];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) *(double2*)(out + 2 * j) = double2{x[2 * j], x[2 * j + 1]};
}

// LDS-only workgroup barrier: global stores stay in flight across it
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MFMA operand of a 16-row block held [row][k] in LDS (pitch PP): lane (fr, fk) takes k = 4 fk .. 4 fk + 3 of row fr
// (element s goes into the s-th of the four K = 4 products; A and B use the same assignment)
struct Op16 { double v[4]; };
__device__ __forceinline__ Op16 op16(const double* __restrict__ T)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    const double2 lo = *(const double2*)(T + fr * PP + 4 * fk), hi = *(const double2*)(T + fr * PP + 4 * fk + 2);
    return Op16{{lo.x, lo.y, hi.x, hi.y}};
}
// ... of the TRANSPOSE of a block held [k][row]
__device__ __forceinline__ Op16 op16_t(const double* __restrict__ T)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    Op16 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o.v[s] = T[(4 * fk + s) * PP + fr];
    return o;
}

// one phase of a compute wave; KB and the wave's rows are compile-time constants, so every acc[][] index is one too
// (as a loop over kb the body stayed rolled once -- the unroll pragma is a hint -- and the accumulators went to
// scratch memory: 150 us per block instead of 23)
template <int W, int kb, bool ACC>
__device__ __forceinline__ void diag_phase(v4d (&acc)[3][NSB], const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld)
{
    constexpr int ROWS[3] = {W == 0 ? 0 : (W == 1 ? 1 : 2), W == 0 ? 7 : (W == 1 ? 6 : 5), W == 0 ? -1 : (W == 1 ? 3 : 4)};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    const double* xd = L.XD + (kb & 1) * 16 * PP;
    const double* pa = L.PA;
    double* pa_next = L.PA;                            // (read before barrier M, rewritten after it)
    DG_STAMP(kb, 0);
    // ---- panel(kb): S(P,kb) <- S(P,kb) X_kb^T; the diagonal sub-tile comes back as L_kb
    if constexpr (ACC) {
        // ... by substitution against L_kb (the pivot wave left it in DG): lane l takes row l & 15 of the wave's (l >> 4)-th
        // sub-tile row, published column -> scaled column; the sub-tiles come back into the accumulators from there
        const int sp = lane >> 4;
        const int P = sp == 0 ? ROWS[0] : (sp == 1 ? ROWS[1] : (sp == 2 ? ROWS[2] : -1));
        if (P >= 0 && P != kb)
            subst16_row(pa + (16 * P + fr) * PP, L.DG + (kb & 1) * 16 * PP, xd, PP + 1, L.PB + (16 * P + fr) * PP);
        wave_lds_sync();
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            if (ROWS[pp] < 0) continue;
            acc[pp][kb] = get16(ROWS[pp] == kb ? L.DG + (kb & 1) * 16 * PP : L.PB + (16 * ROWS[pp]) * PP);
        }
    } else {
        const Op16 xb = op16(xd);
        Op16 a[3];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
            if (ROWS[pp] >= 0 && ROWS[pp] != kb) a[pp] = op16(pa + (16 * ROWS[pp]) * PP);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
            if (ROWS[pp] >= 0 && ROWS[pp] != kb) acc[pp][kb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
                if (ROWS[pp] >= 0 && ROWS[pp] != kb)
                    acc[pp][kb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[pp].v[s], xb.v[s], acc[pp][kb], 0, 0, 0);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            if (ROWS[pp] < 0) continue;
            if (ROWS[pp] == kb) acc[pp][kb] = get16(L.DG + (kb & 1) * 16 * PP);
            else put16(L.PB + (16 * ROWS[pp]) * PP, acc[pp][kb]);
        }
    }
    DG_STAMP(kb, 1);
    lds_barrier();                                     // M
    DG_STAMP(kb, 2);
    // ---- column kb is final: L's sub-tiles (P >= kb) from the registers; the inverse's (P < kb: S(P,kb) =
    // X(kb,P)^T) read back transposed from the scaled column in LDS so that the stores run along rows, and
    // zeros into the mirror block above the diagonal
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
        const int P = ROWS[pp];
        if (P < 0) continue;
        if (P >= kb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 16 * P + fk + 4 * t, col = 16 * kb + fr;
                if (P > kb || col <= row) Bt[(size_t)row * ld + col] = acc[pp][kb][t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = fk + 4 * t;
                Xt[(size_t)(16 * kb + r) * ld + 16 * P + fr] = L.PB[(16 * P + fr) * PP + r];
                Xt[(size_t)(16 * P + r) * ld + 16 * kb + fr] = 0.0;
            }
        }
    }
    if (kb < NSB - 1) {
        // ---- update(kb): S(P,Q) -= S(P,kb) S(Q,kb)^T (Q > kb; P < kb or P >= Q), S(kb,Q) = -X_kb^T S(Q,kb)^T;
        // the operands: row blocks of the scaled column, and X_kb^T for row kb
        Op16 rb[NSB];
#pragma unroll
        for (int Q = 0; Q < NSB; ++Q) {
            bool need = Q > kb;                      // as B operand
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) need = need || (ROWS[pp] == Q && Q != kb);
            if (need) rb[Q] = op16(L.PB + (16 * Q) * PP);
        }
        Op16 na[3];                                  // -A per owned row
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            const int P = ROWS[pp];
            if (P < 0) continue;
            const Op16 src = (P == kb) ? op16_t(xd) : rb[P];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) na[pp].v[s2] = -src.v[s2];
        }
        // column kb+1 first (published as the next panel), with it the diagonal sub-tile kb+2 for the pivot wave
        // A DIAGONAL sub-tile (P == Q) holds the matrix' diagonal entries, ~1 in B = I + D^1/2 K D^1/2 where the update is
        // ~d K: its four MFMA steps accumulate from zero (du) and are added once -- one rounding at the entry's magnitude
        // per phase instead of four (the pivots' accuracy: tile_mma SYM, profiles/r05_var_accuracy.txt)
        v4d du[3];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) du[pp] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int Q = 0; Q < NSB; ++Q) {
                        const int P = ROWS[pp];
                        if (P < 0 || Q <= kb) continue;
                        if (P == kb + 1 && Q == kb + 1) continue;          // the pivot wave's tile
                        if (!(P == kb || P < kb || P >= Q)) continue;
                        const int which = (Q == kb + 1) ? 0 : ((Q == kb + 2 && P == Q) ? 1 : 2);
                        if (which != pass) continue;
                        if (P == Q) du[pp] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[pp].v[s2], rb[Q].v[s2], du[pp], 0, 0, 0);
                        else acc[pp][Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[pp].v[s2], rb[Q].v[s2], acc[pp][Q], 0, 0, 0);
                    }
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {               // (a wave's diagonal sub-tile of this pass: P == kb + 2 in pass 1, beyond in pass 2)
                const int P = ROWS[pp];
                if (P > kb + 1 && P < NSB && ((P == kb + 2) ? 1 : 2) == pass) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[pp][P][t] += du[pp][t];
                }
            }
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                const int P = ROWS[pp];
                if (P < 0) continue;
                if (pass == 0 && kb + 1 < NSB && !(P == kb + 1)) put16(pa_next + (16 * P) * PP, acc[pp][kb + 1]);
                if (pass == 1 && kb + 2 < NSB && P == kb + 2) put16(L.DG + (kb & 1) * 16 * PP, acc[pp][kb + 2]);
            }
        }
    }
    DG_STAMP(kb, 3);
    lds_barrier();                                     // E
    DG_STAMP(kb, 4);
}

// where the tile's entries come from: its own memory (Bt), or -- the small-N path -- formed on the way in from K and s
struct DiagFromTile {
    gptr_t Bt; int ld;
    __device__ __forceinline__ double operator()(int row, int col) const { return Bt[(size_t)row * ld + col]; }
};

// nph: sub-tile columns (phases) that hold data -- beyond them the tile is its identity padding, whose factor and inverse
// are the identity again (diag_identity_tail writes it); every wave of the workgroup gets the same value
template <int W, class LOAD, bool ACC>
__device__ __forceinline__ void diag_compute(const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld, const LOAD& load, int nph)
{
    // this wave's sub-tile rows (-1 = none): equal update counts
    constexpr int ROWS[3] = {W == 0 ? 0 : (W == 1 ? 1 : 2), W == 0 ? 7 : (W == 1 ? 6 : 5), W == 0 ? -1 : (W == 1 ? 3 : 4)};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    v4d acc[3][NSB];
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int Q = 0; Q < NSB; ++Q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int P = ROWS[pp];
                if (P < 0 || Q > P) { acc[pp][Q][t] = 0.0; continue; }
                const int row = 16 * P + fk + 4 * t, col = 16 * Q + fr;
                const double v = load(row, col);
                acc[pp][Q][t] = (Q < P || col <= row) ? v : 0.0;
            }
    // publish column 0 and the diagonal sub-tile 1 as they are
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
        const int P = ROWS[pp];
        if (P < 0) continue;
        put16(L.PA + (16 * P) * PP, acc[pp][0]);
        if (P == 1) put16(L.DG + 16 * PP, acc[pp][1]);      // (sub-tile 0: the pivot wave fetches it itself)
    }
    DG_STAMP(NSB, 0);
    lds_barrier();
    DG_STAMP(NSB, 1);
    DG_STAMP(NSB, 2);
    lds_barrier();                                  // (the pivot wave factored sub-tile 0 in between)
    DG_STAMP(NSB, 3);

    // (uniform branches around straight-line phases: every index into acc stays a compile-time constant)
    diag_phase<W, 0, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 1) diag_phase<W, 1, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 2) diag_phase<W, 2, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 3) diag_phase<W, 3, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 4) diag_phase<W, 4, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 5) diag_phase<W, 5, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 6) diag_phase<W, 6, ACC>(acc, L, Bt, Xt, ld);
    if (nph > 7) diag_phase<W, 7, ACC>(acc, L, Bt, Xt, ld);
}

template <class LOAD, bool ACC>
__device__ __forceinline__ void diag_pivot(const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld, const LOAD& load, int* __restrict__ info, int slot,
                                           int pivot0, int nph)
{
    DG_STAMP(NSB, 0);
    {   // sub-tile (0,0) straight from memory and factored while the compute waves still fetch theirs
        const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
        v4d c0;
#pragma unroll
        for (int t = 0; t < 4; ++t) c0[t] = load(fk + 4 * t, fr);
        DG_STAMP(NSB, 1);
        base16_regs<ACC>(c0, L.DG, L.XD, Xt, ld, info, slot, pivot0, L.LINE);
    }
    DG_STAMP(NSB, 2);
    lds_barrier();
    lds_barrier();
    DG_STAMP(NSB, 3);
#pragma unroll 1
    for (int kb = 0; kb < nph; ++kb) {
        const double* xd = L.XD + (kb & 1) * 16 * PP;
        const double* pa = L.PA;
        const int n = kb + 1;
        DG_STAMP(kb, 0);
        v4d tt = (v4d){0.0, 0.0, 0.0, 0.0};
        if (kb < NSB - 1) {
            // ---- one step ahead: bring S(kb+1,kb+1) up to date through step kb.  L'^T = X_kb S(kb+1,kb)^T comes
            // out of the MFMA as lane (fr = r, fk) holding L'[r][fk + 4t] -- an operand layout of L' (K index
            // fk + 4t for the t-th product), the same for both sides of L' L'^T: no trip through LDS
            v4d uu = (v4d){0.0, 0.0, 0.0, 0.0};
            if constexpr (ACC) {
                // L' by substitution (subst16_row: the compute wave that owns row n does the same on the same input), a copy
                // of this wave's own; from there as an operand layout of L' for both sides of L' L'^T
                if ((threadIdx.x & 63) < 16)
                    subst16_row(pa + (16 * n + (threadIdx.x & 15)) * PP, L.DG + (kb & 1) * 16 * PP, xd, PP + 1,
                                L.LP + (threadIdx.x & 15) * PP);
                wave_lds_sync();
                const Op16 lp = op16(L.LP);
#pragma unroll
                for (int s = 0; s < 4; ++s) uu = __builtin_amdgcn_mfma_f64_16x16x4f64(lp.v[s], lp.v[s], uu, 0, 0, 0);
            } else {
            const Op16 xa = op16(xd), sb = op16(pa + (16 * n) * PP);
            v4d lt = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) lt = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.v[s], sb.v[s], lt, 0, 0, 0);
            // (L' L'^T from zero, subtracted once: the sub-tile holds diagonal entries -- see the compute waves' du)
#pragma unroll
            for (int s = 0; s < 4; ++s) uu = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[s], lt[s], uu, 0, 0, 0);
            }
            tt = get16(L.DG + (n & 1) * 16 * PP);
#pragma unroll
            for (int t = 0; t < 4; ++t) tt[t] -= uu[t];
        }
        DG_STAMP(kb, 1);
        lds_barrier();                                     // M
        DG_STAMP(kb, 2);
        if (kb < NSB - 1)                                  // ... then factor it, straight from the registers
            base16_regs<ACC>(tt, L.DG + (n & 1) * 16 * PP, L.XD + (n & 1) * 16 * PP, Xt + (size_t)(16 * n) * ld + 16 * n, ld,
                        info, slot, pivot0 + 16 * n, L.LINE);
        DG_STAMP(kb, 3);
        lds_barrier();                                     // E
        DG_STAMP(kb, 4);
    }
}

// potrf + inverse of a 128x128 tile whose entries `load(row, col)` supplies: L (lower) -> Bt, X = L^-1 -> Xt;
// `lds`: DIAG_LDS_DOUBLES doubles.  All 256 threads of the workgroup call it.
// Sub-tile columns kb >= nph of a tile whose data end before them: identity padding in, identity out -- what the phases
// kb >= nph of the full schedule would have stored (they multiply and subtract exact zeros): column kb of L (lower part:
// ones on the diagonal), row block kb of X (zeros left of its diagonal sub-tile, the identity on it) and zeros in the
// mirror block above the diagonal.  All 256 threads.
__device__ __forceinline__ void diag_identity_tail(gptr_t Bt, gptr_t Xt, int ld, int nph)
{
    const int tid = threadIdx.x;
    for (int kb = nph; kb < NSB; ++kb) {
        const int c0 = 16 * kb;
        for (int idx = tid; idx < (128 - c0) * 16; idx += 256) {
            const int row = c0 + idx / 16, col = c0 + idx % 16;
            if (col <= row) Bt[(size_t)row * ld + col] = row == col ? 1.0 : 0.0;
        }
        for (int idx = tid; idx < 16 * (c0 + 16); idx += 256) {
            const int r = idx / (c0 + 16), col = idx % (c0 + 16);
            Xt[(size_t)(c0 + r) * ld + col] = col == c0 + r ? 1.0 : 0.0;
        }
        for (int idx = tid; idx < c0 * 16; idx += 256) {
            const int row = idx / 16, col = c0 + idx % 16;
            Xt[(size_t)row * ld + col] = 0.0;
        }
    }
}

// ACC: panel steps by substitution (subst16_row; `lds` then has DIAG_LDS_DOUBLES_ACC doubles)
template <class LOAD, bool ACC = false>
__device__ __forceinline__ void diag_tile_from(double* __restrict__ lds, const LOAD& load, gptr_t Bt, gptr_t Xt, int ld,
                                               int* __restrict__ info, int slot, int pivot0, int nph = NSB)
{
    const DiagLds L(lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nph = __builtin_amdgcn_readfirstlane(nph < 1 ? 1 : (nph > NSB ? NSB : nph));
    if (wave == 0) diag_compute<0, LOAD, ACC>(L, Bt, Xt, ld, load, nph);
    else if (wave == 1) diag_compute<1, LOAD, ACC>(L, Bt, Xt, ld, load, nph);
    else if (wave == 2) diag_compute<2, LOAD, ACC>(L, Bt, Xt, ld, load, nph);
    else diag_pivot<LOAD, ACC>(L, Bt, Xt, ld, load, info, slot, pivot0, nph);
    if (nph < NSB) diag_identity_tail(Bt, Xt, ld, nph);
}

// ... of the tile at Bt itself
template <bool ACC = false>
__device__ __forceinline__ void diag_tile(double* __restrict__ lds, gptr_t Bt, gptr_t Xt, int ld,
                                          int* __restrict__ info, int slot, int pivot0, int nph = NSB)
{
    diag_tile_from<DiagFromTile, ACC>(lds, DiagFromTile{Bt, ld}, Bt, Xt, ld, info, slot, pivot0, nph);
}

// ---- ACC at the tile level: 16 rows of a panel tile, x L_kk^T = b in place, by ONE wave.
// Blocked by 16 like the tile itself: block column cb of the rows is solved by substitution against the diagonal block
// L_kk[cb,cb] (subst16_row, one lane per row), then taken out of the block columns right of it on the matrix cores,
// b[., cb'] -= x_cb L_kk[cb',cb]^T (the operand blocks of L_kk straight from memory, lane (n, fk) four consecutive k of row
// n) -- LAPACK's blocked trsm with its 16 x 16 solves by substitution.  Every caller (the chain's tile, the side stream's
// panel launch, the two-tile small path) runs this function on whole 16-row blocks, so the result does not depend on who
// computed it.  rows: 16 rows x 128 columns, leading dimension ld; Lkk: the factored diagonal tile (lower part read);
// Xkk: its inverse (diagonal read: the reciprocal pivots); scr: TRSM_SCRATCH doubles of LDS of this wave's own.
#define TRSM_SCRATCH (2 * 16 * PP + 16)
__device__ __forceinline__ void trsm_rows16(double* __restrict__ scr, gptr_t rows, gcptr_t Lkk, gcptr_t Xkk, int ld)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    double* const SA = scr;                  // the block column being solved, [row][PP]
    double* const SL = scr + 16 * PP;        // L_kk[cb,cb], [row][PP]
    double* const SR = scr + 2 * 16 * PP;    // its reciprocal pivots
    v4d acc[NSB];
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[cb][t] = rows[(size_t)(fk + 4 * t) * ld + 16 * cb + fr];
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb) {
        // operands of this step, requested up front: the diagonal block (lane (row, part): four consecutive entries), the
        // reciprocal pivots, and the blocks below it
        gcptr_t Ld = Lkk + (size_t)(16 * cb) * ld + 16 * cb;
        const v2d d0 = *(const GPRN_GLOBAL v2d*)(Ld + (size_t)fr * ld + 4 * fk);
        const v2d d1 = *(const GPRN_GLOBAL v2d*)(Ld + (size_t)fr * ld + 4 * fk + 2);
        const double rv = l < 16 ? Xkk[(size_t)(16 * cb + l) * ld + 16 * cb + l] : 0.0;
        Op16 lb[NSB];
#pragma unroll
        for (int c2 = cb + 1; c2 < NSB; ++c2) {
            gcptr_t Lo = Lkk + (size_t)(16 * c2 + fr) * ld + 16 * cb + 4 * fk;
            const v2d lo = *(const GPRN_GLOBAL v2d*)Lo, hi = *(const GPRN_GLOBAL v2d*)(Lo + 2);
            lb[c2] = Op16{{lo.x, lo.y, hi.x, hi.y}};
        }
        put16(SA, acc[cb]);
        *(v2d*)(SL + fr * PP + 4 * fk) = d0;
        *(v2d*)(SL + fr * PP + 4 * fk + 2) = d1;
        if (l < 16) SR[l] = rv;
        wave_lds_sync();
        if (l < 16) subst16_row(SA + l * PP, SL, SR, 1, SA + l * PP);
        wave_lds_sync();
        // the solved block: to memory (lane (row, part): 32 bytes of its row) and, as the A operand, to the updates
        {
            const v2d x0 = *(const v2d*)(SA + fr * PP + 4 * fk), x1 = *(const v2d*)(SA + fr * PP + 4 * fk + 2);
            *(GPRN_GLOBAL v2d*)(rows + (size_t)fr * ld + 16 * cb + 4 * fk) = x0;
            *(GPRN_GLOBAL v2d*)(rows + (size_t)fr * ld + 16 * cb + 4 * fk + 2) = x1;
#pragma unroll
            for (int c2 = cb + 1; c2 < NSB; ++c2) {
                acc[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0.x, lb[c2].v[0], acc[c2], 0, 0, 0);
                acc[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0.y, lb[c2].v[1], acc[c2], 0, 0, 0);
                acc[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1.x, lb[c2].v[2], acc[c2], 0, 0, 0);
                acc[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1.y, lb[c2].v[3], acc[c2], 0, 0, 0);
            }
        }
        wave_lds_sync();                     // (SA, SL are rewritten by the next step)
    }
}

