// Blocked right-looking fp64 Cholesky that builds the inverse factor in the
// same sweep:  B = L L^T  and  X = L^-1,  batched over latent GPs.
//
// Replaces, per latent GP and per ELBOaux call, the reference's
//   np.linalg.solve(diag(1/d)+K, K) + K @ (...)      meanfield.py:771,850
//   cholesky(Sigma)                                  meanfield.py:1087,1090
//   cho_solve(L_K, Sigma) for a trace                meanfield.py:1041,1051
// with POTRF(B) + TRTRI(L_B) on B = I + D^1/2 K D^1/2, and at
// setup the reference's _cholNugget(K) (meanfield.py:71-89,621-622).
//
// Step k of T = ld/128 (tile row/col k):
//   diag   : L_kk = chol(B_kk), X_kk = L_kk^-1            one workgroup, LDS
//   panel  : L_ik = B_ik X_kk^T (i>k);  X_kc = X_kk R_kc (c<k)   tile GEMMs, K=128
//   update : B_ij -= L_ik L_jk^T (i>=j>k);  R_ic -= L_ik X_kc (i>k, c<=k)
// where R (the running right-hand side of L X = I) lives in X's own tiles and
// tile (i,c) is first written, not accumulated, at step k == c.  Per step the
// update touches (T-1-k)(T-k)/2 + (T-1-k)(k+1) tiles -- roughly constant until
// the tail, unlike POTRF alone.  Flops: N^3/3 + N^3/3.
#include "gprn_internal.h"
#include "dag.h"
#include "tile_mma.h"
#include "vecops.h"

#include <math.h>
#include <time.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>

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
    double y = __builtin_amdgcn_rsq(x);
    const double hx = -0.5 * x;
    y = y * fma(hx * y, y, 1.5);
    y = y * fma(hx * y, y, 1.5);      // second step: seed accuracy is not documented for gfx950 (and dropping
                                      // it does not shorten base16: 7537 vs 7701 cycles, _probe/base16_bench.hip)
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
#ifdef BASE16_STAMPS     // _probe/base16_bench.hip: shader-clock stamps inside one call, after `dep` is available
__device__ long long b16_stamps[4][8];
#define B16_STAMP(R, i, dep) do { asm volatile("" :: "v"(dep)); b16_stamps[R][i] = clock64(); } while (0)
#else
#define B16_STAMP(R, i, dep) do {} while (0)
#endif
// c: the block in C layout (what lies above the diagonal is ignored); L -> St (lower), X -> xd and Xg
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
#pragma unroll
        for (int n = 3; n >= 0; --n) {
            double u = 0.0;
#pragma unroll
            for (int k = n + 1; k < 4; ++k) u = fma(xr[k], L[k][n], u);
            xr[n] = (fk == n) ? inv[n] : -u * inv[n];
            w = fma(sp[n], xr[n], w);
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
__device__ __forceinline__ void base16(double* __restrict__ St /* pitch PP */,
                                       double* __restrict__ xd, gptr_t Xg, int ld,
                                       int* info, int slot, int pivot0,
                                       double* __restrict__ line /* 64 doubles of LDS */)
{
    const int l = threadIdx.x & 63, fr = l & 15, fk = l >> 4;
    v4d c;
#pragma unroll
    for (int t = 0; t < 4; ++t) c[t] = St[(fk + 4 * t) * PP + fr];
    base16_regs(c, St, xd, Xg, ld, info, slot, pivot0, line);
}

// The form of rounds 1-2, kept for _probe/base16_bench.hip: lane (r = l&15, g = l>>4) owns columns

// 4g..4g+3 of row r; strict upper = transposed right-hand side, as in S.
//
// Four pivots per round: the 4x4 diagonal block of the round is fetched with v_readlane
// (10 values) and factored + inverted analytically by every lane at once (uniform data, no
// cross-lane step inside the block); the lanes that own the block's four columns then apply
// the panel solve to their row, publish the four scaled values through one LDS line, and all
// lanes apply the rank-4 update.  One LDS round trip and four reciprocal-square-root chains
// per four pivots -- the serial cost per pivot drops about threefold against the
// one-pivot-per-step form (539 cycles per pivot measured for that one).
__device__ __forceinline__ void base16_lanes(double* __restrict__ St /* pitch PP */,
                                       double* __restrict__ xd, gptr_t Xg, int ld,
                                       int* info, int slot, int pivot0,
                                       double* __restrict__ line /* 64 doubles of LDS */)
{
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * g + j;
        a[j] = (b <= r) ? St[r * PP + b] : 0.0;
    }
    int bad_at = 0;
#pragma unroll
    for (int R = 0; R < 4; ++R) {
        const int k0 = 4 * R;
        // ---- the 4x4 diagonal block (rows k0..k0+3 of column group R), lower part, to every lane
        double d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m <= i; ++m) d[i][m] = readlane_f64(a[m], k0 + i + 16 * R);
        // ---- its Cholesky factor L (lower) and X = L^-1, all lanes redundantly
        double inv[4], L[4][4], X[4][4];
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            X[i][i] = inv[i];
#pragma unroll
            for (int c = i - 1; c >= 0; --c) {          // X[i][c] = -inv_i * sum_{m=c}^{i-1} L[i][m] X[m][c]
                double u = 0.0;
#pragma unroll
                for (int m = c; m < i; ++m) u = fma(L[i][m], X[m][c], u);
                X[i][c] = -u * inv[i];
            }
        }
        // ---- panel: the owners of columns k0..k0+3 scale their row, w = raw X^T; rows inside the
        // block take row i of X^T instead (first touch of the inverse's rows) and store L / X^T
        double w[4];
        const bool owner = (g == R), inside = (r >= k0 && r < k0 + 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double u = 0.0;
#pragma unroll
            for (int n = 0; n <= m; ++n) u = fma(a[n], X[m][n], u);
            w[m] = u;
        }
        if (inside) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r == k0 + i) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) w[m] = (m >= i) ? X[m][i] : 0.0;
                }
        }
        if (owner) {
#pragma unroll
            for (int m = 0; m < 4; ++m) line[4 * r + m] = w[m];
            // what stays in the registers of the block's columns
            if (inside) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r == k0 + i) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) a[m] = (m <= i) ? L[i][m] : X[m][i];
                    }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = w[m];
            }
        }
        wave_lds_sync();                        // same wave, in-order LDS: the reads below see the line
        if (R < 3) {
            // ---- rank-4 update of the columns to the right: a[j] -= sum_m W[r][m] W[b][m]
            double wr[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) wr[m] = line[4 * r + m];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = 4 * g + j;
                double u = a[j];
#pragma unroll
                for (int m = 0; m < 4; ++m) u = fma(-wr[m], line[4 * b + m], u);
                if (g > R && (r < k0 + 4 || r >= b)) a[j] = u;
            }
        }
        wave_lds_sync();
    }
    if (bad_at && l == 0 && info[slot] == 0) info[slot] = pivot0 + bad_at;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * g + j;
        if (b <= r) St[r * PP + b] = a[j];
        if (b > r) {
            xd[b * PP + r] = a[j];  xd[r * PP + b] = 0.0;
            Xg[(size_t)b * ld + r] = a[j];  Xg[(size_t)r * ld + b] = 0.0;
        } else if (b == r) {
            const double x = 1.0 / a[j];
            xd[r * PP + r] = x;
            Xg[(size_t)r * ld + r] = x;
        }
    }
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
#ifdef DIAG_STAMPS        // _probe/diag_bench.hip: shader-clock stamps per wave, phase and point
__device__ long long diag_stamps[4][NSB + 1][6];
#define DG_STAMP(kb, i) do { if ((threadIdx.x & 63) == 0) diag_stamps[threadIdx.x >> 6][kb][i] = clock64(); } while (0)
#else
#define DG_STAMP(kb, i) do {} while (0)
#endif
// 46.6 KB: the kernel fits on a CU beside two bulk-update workgroups (2 x (41 + 15) KB of the CU's 160) or one with the
// small-batch pad.  (Round 1-2 form: 67 KB with the published column double-buffered -- it is read before barrier M and
// rewritten after it, one buffer does -- and a scratch tile the pivot wave no longer needs.)
#define DIAG_LDS_DOUBLES (128 * PP + 128 * PP + 2 * 16 * PP + 2 * 16 * PP + 64)

struct DiagLds {
    double *PA, *PB, *DG, *XD, *LINE;
    __device__ explicit DiagLds(double* lds)
        : PA(lds),                       // published column panel (before its scaling)        128 x PP
          PB(PA + 128 * PP),             // current column after scaling by X_kb^T              128 x PP
          DG(PB + 128 * PP),             // diagonal sub-tiles for / from the pivot wave (by parity)
          XD(DG + 2 * 16 * PP),          // X_kb by parity
          LINE(XD + 2 * 16 * PP) {}
};

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
template <int W, int kb>
__device__ __forceinline__ void diag_phase(v4d (&acc)[3][NSB], const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld)
{
    constexpr int ROWS[3] = {W == 0 ? 0 : (W == 1 ? 1 : 2), W == 0 ? 7 : (W == 1 ? 6 : 5), W == 0 ? -1 : (W == 1 ? 3 : 4)};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    const double* xd = L.XD + (kb & 1) * 16 * PP;
    const double* pa = L.PA;
    double* pa_next = L.PA;                            // (read before barrier M, rewritten after it)
    DG_STAMP(kb, 0);
    // ---- panel(kb): S(P,kb) <- S(P,kb) X_kb^T; the diagonal sub-tile comes back as L_kb
    {
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
                        acc[pp][Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[pp].v[s2], rb[Q].v[s2], acc[pp][Q], 0, 0, 0);
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

template <int W>
__device__ __forceinline__ void diag_compute(const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld, const double* __restrict__ img)
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
                const double v = img ? img[(P * (P + 1) / 2 + Q) * 256 + (fk + 4 * t) * 16 + fr] : Bt[(size_t)row * ld + col];
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

    diag_phase<W, 0>(acc, L, Bt, Xt, ld); diag_phase<W, 1>(acc, L, Bt, Xt, ld);
    diag_phase<W, 2>(acc, L, Bt, Xt, ld); diag_phase<W, 3>(acc, L, Bt, Xt, ld);
    diag_phase<W, 4>(acc, L, Bt, Xt, ld); diag_phase<W, 5>(acc, L, Bt, Xt, ld);
    diag_phase<W, 6>(acc, L, Bt, Xt, ld); diag_phase<W, 7>(acc, L, Bt, Xt, ld);
}

__device__ __forceinline__ void diag_pivot(const DiagLds& L, gptr_t Bt, gptr_t Xt, int ld, int* __restrict__ info, int slot,
                                           int pivot0, const double* __restrict__ img)
{
    DG_STAMP(NSB, 0);
    {   // sub-tile (0,0) straight from memory and factored while the compute waves still fetch theirs
        const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
        v4d c0;
#pragma unroll
        for (int t = 0; t < 4; ++t) c0[t] = img ? img[(fk + 4 * t) * 16 + fr] : Bt[(size_t)(fk + 4 * t) * ld + fr];
        DG_STAMP(NSB, 1);
        base16_regs(c0, L.DG, L.XD, Xt, ld, info, slot, pivot0, L.LINE);
    }
    DG_STAMP(NSB, 2);
    lds_barrier();
    lds_barrier();
    DG_STAMP(NSB, 3);
#pragma unroll 1
    for (int kb = 0; kb < NSB; ++kb) {
        const double* xd = L.XD + (kb & 1) * 16 * PP;
        const double* pa = L.PA;
        const int n = kb + 1;
        DG_STAMP(kb, 0);
        v4d tt = (v4d){0.0, 0.0, 0.0, 0.0};
        if (kb < NSB - 1) {
            // ---- one step ahead: bring S(kb+1,kb+1) up to date through step kb.  L'^T = X_kb S(kb+1,kb)^T comes
            // out of the MFMA as lane (fr = r, fk) holding L'[r][fk + 4t] -- an operand layout of L' (K index
            // fk + 4t for the t-th product), the same for both sides of L' L'^T: no trip through LDS
            const Op16 xa = op16(xd), sb = op16(pa + (16 * n) * PP);
            v4d lt = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) lt = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.v[s], sb.v[s], lt, 0, 0, 0);
            tt = get16(L.DG + (n & 1) * 16 * PP);
#pragma unroll
            for (int s = 0; s < 4; ++s) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(-lt[s], lt[s], tt, 0, 0, 0);
        }
        DG_STAMP(kb, 1);
        lds_barrier();                                     // M
        DG_STAMP(kb, 2);
        if (kb < NSB - 1)                                  // ... then factor it, straight from the registers
            base16_regs(tt, L.DG + (n & 1) * 16 * PP, L.XD + (n & 1) * 16 * PP, Xt + (size_t)(16 * n) * ld + 16 * n, ld,
                        info, slot, pivot0 + 16 * n, L.LINE);
        DG_STAMP(kb, 3);
        lds_barrier();                                     // E
        DG_STAMP(kb, 4);
    }
}

// potrf + inverse of the 128x128 tile at Bt (-> L, lower) with X = L^-1 -> Xt; `lds`: DIAG_LDS_DOUBLES doubles.
// All 256 threads of the workgroup call it.
// img: the tile's lower 16 x 16 blocks in LDS instead of at Bt (block (P, Q) at img + (P (P + 1) / 2 + Q) * 256, row-major),
// or null.
__device__ __forceinline__ void diag_tile(double* __restrict__ lds, gptr_t Bt, gptr_t Xt, int ld,
                                          int* __restrict__ info, int slot, int pivot0,
                                          const double* __restrict__ img = nullptr)
{
    const DiagLds L(lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) diag_compute<0>(L, Bt, Xt, ld, img);
    else if (wave == 1) diag_compute<1>(L, Bt, Xt, ld, img);
    else if (wave == 2) diag_compute<2>(L, Bt, Xt, ld, img);
    else diag_pivot(L, Bt, Xt, ld, info, slot, pivot0, img);
}

// PTRS: the two pointers per matrix come as kernel arguments (launch_diag), else from the table
// (one wave per SIMD: the register-resident tile needs ~290 VGPRs per lane; without the second bound the compiler sizes
// the allocation for the three workgroups per CU the LDS would allow and spills)
template <bool ARGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_diag_block(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int kblk, int* __restrict__ info,
                  unsigned* sig_slot, unsigned sig_value, const unsigned* wait_flag, unsigned wait_value,
                  unsigned* wait_timed_out)
{
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
    CHAIN_PRIO();
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[0] = __builtin_amdgcn_s_memrealtime();
    await_flag(wait_flag, wait_value, wait_timed_out);
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[1] = __builtin_amdgcn_s_memrealtime();
    const int slot = blockIdx.x;
    const size_t off = ((size_t)kblk * GPRN_TILE) * ld + (size_t)kblk * GPRN_TILE;
    double* const Bm = ARGS ? pa.p[slot][0] : ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = ARGS ? pa.p[slot][1] : ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    diag_tile(lds, (gptr_t)(Bm + off), (gptr_t)(Xm + off), ld, info, slot, kblk * GPRN_TILE);
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[2] = __builtin_amdgcn_s_memrealtime();
    signal_done(sig_slot, sig_value, nullptr, 0, nullptr);
}

// The same kernel as a node of the dataflow schedule (queue.hip): it polls its own node and tells its successors.
template <bool ARGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_diag_block_q(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int kblk, int* __restrict__ info,
                    QueueCtl q, unsigned qop)
{
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
    const int slot = blockIdx.x;
    CHAIN_PRIO();
    const unsigned long long t0 = q.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    q_await(q, slot, qop);
    const unsigned long long t1 = q.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const size_t off = ((size_t)kblk * GPRN_TILE) * ld + (size_t)kblk * GPRN_TILE;
    double* const Bm = ARGS ? pa.p[slot][0] : ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = ARGS ? pa.p[slot][1] : ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    diag_tile(lds, (gptr_t)(Bm + off), (gptr_t)(Xm + off), ld, info, slot, kblk * GPRN_TILE);
    q_complete(q, slot, qop, false);
    if (q.trace && threadIdx.x == 0)
        q_trace(q, (unsigned long long)q_entry(slot, 0, qop) | (0xffffull << 32), t0, t1, __builtin_amdgcn_s_memrealtime());
}

int launch_diag_q(gprn_ctx* c, double** d_ptrs, int nbatch, int ld, int kblk, int* d_info, hipStream_t stream,
                  const QueueCtl& q, unsigned op)
{
    prof_begin(c, GPRN_T_DIAG, stream);
    PtrArgs pa;
    pa.stamps = nullptr;
    if (tab_rows(c, d_ptrs, nbatch, &pa))
        hipLaunchKernelGGL(k_diag_block_q<true>, dim3(nbatch), dim3(256), 0, stream, (double* const*)d_ptrs, pa, ld, kblk, d_info, q, op);
    else
        hipLaunchKernelGGL(k_diag_block_q<false>, dim3(nbatch), dim3(256), 0, stream, (double* const*)d_ptrs, pa, ld, kblk, d_info, q, op);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

void tab_note(gprn_ctx* c, double** d_tab, double* const* rows, size_t count)
{
    for (auto& e : c->tab_host)
        if (e.first == d_tab) { e.second.assign(rows, rows + count); return; }
    c->tab_host.emplace_back(d_tab, std::vector<double*>(rows, rows + count));
}

void tab_forget(gprn_ctx* c, double** d_tab)
{
    if (!d_tab) { c->tab_host.clear(); return; }
    for (size_t i = 0; i < c->tab_host.size(); ++i)
        if (c->tab_host[i].first == d_tab) { c->tab_host.erase(c->tab_host.begin() + i); return; }
}

bool tab_rows(gprn_ctx* c, double** d_ptrs, int nbatch, PtrArgs* out)
{
    static int on = -1;                            // GPRN_ARG_PTRS=0: always the table (experiments)
    if (on < 0) { const char* e = getenv("GPRN_ARG_PTRS"); on = e ? atoi(e) : 1; }
    if (!on || nbatch > GPRN_ARG_SLOTS) return false;
    for (const auto& e : c->tab_host) {
        if (d_ptrs < e.first || d_ptrs >= e.first + e.second.size()) continue;
        const size_t first = (size_t)(d_ptrs - e.first);
        if (first % GPRN_NBUF || first + (size_t)nbatch * GPRN_NBUF > e.second.size()) return false;
        for (int b = 0; b < nbatch; ++b) {
            out->p[b][0] = e.second[first + (size_t)b * GPRN_NBUF + BUF_B];
            out->p[b][1] = e.second[first + (size_t)b * GPRN_NBUF + BUF_X];
        }
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ chain
// The latency chain of one factorisation as ONE persistent workgroup per matrix: for every tile step k
//     diag(k)  ->  L_{k+1,k} = B_{k+1,k} X_kk^T  ->  B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T
// back to back on a CU of its own (it asks for so much LDS that nothing else fits beside it), instead of three
// dependent launches per step that share their CUs with the bulk updates: no launch, dispatch or flag latency
// inside the chain, no co-resident MFMA waves holding the SIMD's FP64 units while the pivot chain runs.
// It talks to the other streams through the same flags as the launch schedule (factor_invert_split):
//   raises  F_DIAG(k)   when L_kk, X_kk are in memory   (stream3 starts the panel of step k)
//           F_MINIL(k)  when L_{k+1,k} is                (stream3's in-panel updates of step k)
//   waits   F_INNER(k-1) before it reads B_{k+1,k}, B_{k+1,k+1}  (stream3's in-panel update of step k-1)
// flags: (step or panel) * kinds * 2 + kind * 2 + 1 words into `sig`; a flag is up when it holds >= epoch.
#define CHAIN_MMA_DOUBLES (2 * 16 * (128 + 128 + 32))
#define CHAIN_LDS_DOUBLES (CHAIN_MMA_DOUBLES > DIAG_LDS_DOUBLES ? CHAIN_MMA_DOUBLES : DIAG_LDS_DOUBLES)

// One flag per (tile step, kind) serves the whole batch: the chain workgroups of all matrices count in on the
// word in front of it and the last one raises it (as the workgroups of one launch do in signal_done).
__device__ __forceinline__ void chain_publish(unsigned* flag, unsigned epoch)
{
    // every wave's stores have left the CU, then one release for the workgroup, then the count / the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* const count = flag - 1;
        if (atomicAdd(count, 1u) + 1 == gridDim.x) {
            atomicExch(count, 0u);
            __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256)
void k_chain(double* const* __restrict__ ptrs, int ld, int T, int outer, int* __restrict__ info,
             unsigned* sig, int kinds, int f_diag, int f_minil, int f_inner, int f_first, unsigned epoch,
             unsigned* timed_out, unsigned long long* stamps /* development aid: 8 per tile step, or null */)
{
#define STAMP(i) do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[(size_t)k * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // [0, CHAIN_LDS_DOUBLES): the diagonal block's buffers / the tile products' stages; behind them the next
    // diagonal tile as the update leaves it (36 lower blocks): 144 KiB in all -- nothing else fits on this CU
    __shared__ __attribute__((aligned(16))) double lds[CHAIN_LDS_DOUBLES + 36 * 256];
    double* const img = lds + CHAIN_LDS_DOUBLES;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x;
    double* const Bm = ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    auto flag = [&](int idx, int kind) { return sig + ((size_t)idx * kinds + kind) * 2 + 1; };
    for (int k = 0; k < T; ++k) {
        const size_t dk = ((size_t)k * GPRN_TILE) * ld + (size_t)k * GPRN_TILE;
        STAMP(0);
        diag_tile(lds, (gptr_t)(Bm + dk), (gptr_t)(Xm + dk), ld, info, slot, k * GPRN_TILE, k > 0 ? img : nullptr);
        STAMP(1);
        chain_publish(flag(k, f_diag), epoch);
        STAMP(2);
        if (k + 1 == T) break;
        // L_{k+1,k} = B_{k+1,k} X_kk^T, in place; rows split over the four waves (each meets the same share
        // of X_kk's zero half)
        if (k > 0) await_flag(flag(k - 1, f_inner), epoch, timed_out);
        else __syncthreads();
        STAMP(3);
        const size_t sub = dk + (size_t)GPRN_TILE * ld;          // tile (k+1, k)
        tile_mma<128, 128, 4, 1, 1, false>(lds, Bm + sub, Xm + dk, (gptr_t)(Bm + sub), ld, 0, 0, CM_SET,
                                            GPRN_TILE, 0, 0);
        STAMP(4);
        chain_publish(flag(k, f_minil), epoch);
        STAMP(5);
        // B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T (lower blocks); like every tile the chain touches it is kept
        // up to date by the in-panel updates alone (ensure_tasks), so there is nothing else to wait for
        // every wave its own copy (its blocks are compile-time constants there); the result stays in LDS for
        // the next diagonal block -- nobody else reads this tile before it is factored
        switch (wave_id) {
#define U_OF(W) case W: tile_mma<128, 128, 4, 1, 0, true, W>(lds, Bm + sub, Bm + sub, (gptr_t)(Bm + sub + GPRN_TILE), ld, \
                                                             0, 0, CM_SUB, GPRN_TILE, 0, 0, img); break;
        U_OF(0) U_OF(1) U_OF(2) default: U_OF(3)
#undef U_OF
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
    }
#undef STAMP
}

// GPRN_CHAIN=2: every diagonal block of a matrix by ONE resident workgroup -- no dispatch and no search for a CU
// between tile steps; it waits in-kernel for the flag of the previous step's update (F_U) and raises F_DIAG itself.
// The step's two products stay launches of their own on the fourth stream (the two-stream form of the chain).
template <bool ARGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_diag_chain(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int T, int* __restrict__ info,
                  unsigned* sig, int kinds, int f_diag, int f_u, unsigned epoch, unsigned* timed_out)
{
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
    const int slot = blockIdx.x;
    double* const Bm = ARGS ? pa.p[slot][0] : ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = ARGS ? pa.p[slot][1] : ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    for (int k = 0; k < T; ++k) {
        if (k > 0) await_flag(sig + ((size_t)(k - 1) * kinds + f_u) * 2 + 1, epoch, timed_out);
        const size_t dk = ((size_t)k * GPRN_TILE) * ld + (size_t)k * GPRN_TILE;
        diag_tile(lds, (gptr_t)(Bm + dk), (gptr_t)(Xm + dk), ld, info, slot, k * GPRN_TILE);
        chain_publish(sig + ((size_t)k * kinds + f_diag) * 2 + 1, epoch);
    }
}

// GPRN_STEP_STAMPS=2: a one-thread kernel between stream3's launches of a tile step writes the clock too (it costs the
// stream 3-4 us each: a probe of where stream3's time goes, not of how long the step takes)
__global__ void k_stamp(unsigned long long* at) { if (threadIdx.x == 0) *at = __builtin_amdgcn_s_memrealtime(); }

// GPRN_STEP_STAMPS: where chain kernel `which` (0 diag, 1 L, 2 U) of tile step k of the running factorisation puts its
// three clock stamps, or null
unsigned long long* step_stamp_ptr(gprn_ctx* c, int k, int which)
{
    if (!c->d_step_stamps || c->step_stamps_n < 1 || k >= c->step_stamps_T) return nullptr;
    const int ph = (c->step_stamps_n - 1) & 7;
    return c->d_step_stamps + (((size_t)ph * c->step_stamps_T + k) * 3 + which) * 3;
}

int launch_diag(gprn_ctx* c, double** d_ptrs, int nbatch, int ld, int kblk, int* d_info, hipStream_t stream,
                Signal sig, Await aw)
{
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_DIAG, stream);
    PtrArgs pa;
    pa.stamps = nullptr;
    // Little work in total (batch x tiles <= GPRN_DIAG_PAD_MAX, the latency schedule's problems): most CUs are idle, and
    // with GPRN_DIAG_PAD_KB of unused dynamic LDS (default: all a workgroup may have) the workgroup only lands on a CU
    // that runs nothing else which uses LDS -- no co-resident MFMA waves of the side stream's tile kernels on its SIMDs.
    // Config 2 (N = 2048, one matrix per phase): 686 -> 741 sweeps/s.  On a loaded device it waits for such a CU as long
    // as the neighbours would have cost (config 3 with the pad in the node phase: 109.8 vs 110.1), hence the limit.
    static int diag_pad_kb = -1, diag_pad_max = -1;
    if (diag_pad_kb < 0) { const char* e = getenv("GPRN_DIAG_PAD_KB"); diag_pad_kb = e ? atoi(e) : 113; }
    if (diag_pad_max < 0) { const char* e = getenv("GPRN_DIAG_PAD_MAX"); diag_pad_max = e ? atoi(e) : 32; }
    size_t dyn = 0;
    if (diag_pad_kb > 0 && nbatch * c->T <= diag_pad_max)
        dyn = std::min<size_t>((size_t)diag_pad_kb * 1024, lds_limit(c->device) - DIAG_LDS_DOUBLES * sizeof(double));
    pa.stamps = step_stamp_ptr(c, kblk, 0);
    if (tab_rows(c, d_ptrs, nbatch, &pa))
        hipLaunchKernelGGL(k_diag_block<true>, dim3(nbatch), dim3(256), dyn, stream, (double* const*)d_ptrs, pa, ld, kblk,
                           d_info, sig.slot, sig.value, aw.flag, aw.value, aw.timed_out);
    else
        hipLaunchKernelGGL(k_diag_block<false>, dim3(nbatch), dim3(256), dyn, stream, (double* const*)d_ptrs, pa, ld, kblk,
                           d_info, sig.slot, sig.value, aw.flag, aw.value, aw.timed_out);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ------------------------------------------------------------ task lists
static inline int64_t toff(int ti, int tj, int ld) {
    return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE;
}

// Two-level blocking.  Tile steps are grouped into outer panels of GPRN_OUTER tiles.
// Inside a panel (tile step k in [k0,k1)) only what the panel itself needs is
// updated right away, with K = 128:
//     B_ij -= L_ik L_jk^T   j in (k, k1),  i >= j           (all rows, panel columns)
//     R_ic -= L_ik X_kc     i in (k, k1),  c <= k           (panel rows, all columns)
// and everything outside is updated once per panel with K = (k1-k0)*128 -- each
// trailing tile is then read and written once per 512 columns of L instead of
// once per 128 (16 -> 64 flop per HBM byte: MFMA-bound instead of HBM-bound):
//     B_ij -= L[i,k0:k1] L[j,k0:k1]^T          i >= j >= k1
//     R_ic -= L[i,k0:k1] X[k0:k1,c]            i >= k1, c < k0
//     R_ic  = -L[i,c:k1] X[c:k1,c]             i >= k1, k0 <= c < k1   (first touch)
// The outer update is split into the part the next panel needs ("next": its
// columns of B, its rows of R) and the rest, so the next panel's latency chain
// can run while the rest streams on a second HIP stream.
// GPRN_SCHED=1: every in-panel launch on the chain stream; default (2): split schedule, see
// factor_invert_split
static int sched_mode()
{
    static int mode = 0;
    if (!mode) { const char* e = getenv("GPRN_SCHED"); mode = e && atoi(e) == 1 ? 1 : 2; }
    return mode;
}
static bool split_sched() { return sched_mode() >= 2; }

int ensure_tasks(gprn_ctx* c)
{
    const int T = c->T, ld = c->ld;
    if (c->tasks_T == T && c->d_tasks) return GPRN_OK;
    std::vector<TileTask>& v = c->h_tasks;
    v.clear();
    static int lower_diag = -1;                    // GPRN_LOWER_DIAG=0: diagonal tiles updated in full, as in rounds 1-2
    if (lower_diag < 0) { const char* e = getenv("GPRN_LOWER_DIAG"); lower_diag = e ? atoi(e) : 1; }
    static int outer_big = 0;                      // GPRN_OUTER_TILES overrides (experiments)
    if (!outer_big) { const char* e = getenv("GPRN_OUTER_TILES"); outer_big = e && atoi(e) > 0 ? atoi(e) : GPRN_OUTER; }
    for (int set = 0; set < 2; ++set) {
    static int outer_small = 0;
    if (!outer_small) { const char* e = getenv("GPRN_OUTER_SMALL"); outer_small = e && atoi(e) > 0 ? atoi(e) : GPRN_OUTER_SMALL; }
    const int outer = set ? std::max(outer_big, outer_small) : outer_big;
    std::vector<gprn_ctx::StepRange>& steps = c->steps[set];
    std::vector<gprn_ctx::OuterRange>& outers = c->outers[set];
    steps.assign(T, gprn_ctx::StepRange{0, 0, 0, 0, 0, 0});
    outers.clear();
    for (int k0 = 0; k0 < T; k0 += outer) {
        const int k1 = std::min(T, k0 + outer);
        for (int k = k0; k < k1; ++k) {
            gprn_ctx::StepRange& s = steps[k];
            s.panel0 = v.size();
            for (int i = k + 1; i < T; ++i)            // L_ik = B_ik X_kk^T   (in place)
                v.push_back(TileTask{toff(i, k, ld), toff(i, k, ld), toff(k, k, ld), GPRN_TILE,
                                     BUF_B, BUF_B, BUF_X, tile_modes(CM_SET, 0, 0)});
            s.npanel_l = v.size() - s.panel0;
            for (int cc = 0; cc < k; ++cc)             // X_kc = X_kk R_kc     (in place)
                v.push_back(TileTask{toff(k, cc, ld), toff(k, k, ld), toff(k, cc, ld), GPRN_TILE,
                                     BUF_X, BUF_X, BUF_X, tile_modes(CM_SET, 0, 1)});
            s.npanel = v.size() - s.panel0;
            s.upd0 = v.size();
            // columns of the panel right of step k; and of the NEXT panel its diagonal and sub-diagonal tiles
            // (j,j), (j+1,j), k1 <= j < n1 -- the tiles the chain works on there: kept up to date step by
            // step (K = 128), so that the chain never waits for a K = 512 update of the outer panel
            // Order: the chain's own update (k+1,k+1) first, then the two tiles the chain's NEXT step touches --
            // (k+2,k+1), which becomes L_{k+2,k+1}, and (k+2,k+2), which its update writes -- then everything else: the
            // first `ncol1` tasks are what has to be done before the chain may go on (factor_invert_split launches them
            // on their own where the others must wait for an outer update).
            auto is_crit = [&](int i, int j) { return (i == k + 1 && j == k + 1) || (i == k + 2 && (j == k + 1 || j == k + 2)); };
            for (int pass = 0; pass < 2; ++pass) {
                for (int j = k + 1; j < std::min(T, k1 + outer); ++j)
                    for (int i = j; i < (j < k1 ? T : std::min(T, j + 2)); ++i) {
                        if (is_crit(i, j) != (pass == 0)) continue;
                        v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                             BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0, lower_diag && i == j)});
                    }
                if (pass == 0) s.ncol1 = v.size() - s.upd0;
            }
            for (int i = k + 1; i < k1; ++i)
                for (int cc = 0; cc <= k; ++cc)
                    v.push_back(TileTask{toff(i, cc, ld), toff(i, k, ld), toff(k, cc, ld), GPRN_TILE,
                                         BUF_X, BUF_B, BUF_X,
                                         tile_modes(cc == k ? CM_SETNEG : CM_SUB, 0, 1)});
            s.nupd = v.size() - s.upd0;
        }
        // ---- the same steps' updates in left-looking form (throughput set).  Step k, behind its panel products:
        //   column k+1 of the panel:  B_{i,k+1} -= L[i, k0..k] L[k+1, k0..k]^T   (i >= k+3),  K = 128 (k + 1 - k0)
        //   row k+1 of the inverse's right-hand side:  R_{k+1,c} (-)= L[k+1, ..k] X[..k, c]
        //   the diagonal and sub-diagonal tiles to the right (this panel's and the next one's): column k alone, as before.
        // Every in-panel tile is read and written once per panel instead of up to three times, with K up to 384 instead of
        // 128 -- and the panel's columns are needed ONE PER STEP: the previous panel's K = 512 update of column k+1 has to be
        // there at step k, not all of them at the panel's first step ("next" in groups: grp0 / ngrp below).
        if (set == 0) {
            if (c->lsteps.size() != (size_t)T) c->lsteps.assign(T, gprn_ctx::LStep{0, 0, 0});
            for (int k = k0; k < k1; ++k) {
                gprn_ctx::LStep& ls = c->lsteps[k];
                ls.u0 = v.size();
                const int jc = k + 1, kl = (k + 1 - k0) * GPRN_TILE;
                auto is_crit = [&](int i, int j) { return i == k + 2 && (j == k + 1 || j == k + 2); };
                // the band next to the diagonal -- (j,j), (j+1,j) -- stays right-looking, column k alone (K = 128) at every
                // step, inside the panel as in the next one: the two tiles the chain's next step touches are among them, and
                // with the whole panel's K they took three times as long at the panel's third step (107.6 vs 110.6 sweeps/s)
                for (int pass = 0; pass < 2; ++pass) {
                    for (int j = k + 1; j < std::min(T, k1 + outer); ++j)
                        for (int i = j; i < std::min(T, j + 2); ++i) {
                            if (i == k + 1 && j == k + 1) continue;            // the chain's own update of this step
                            if (is_crit(i, j) != (pass == 0)) continue;
                            v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                                 BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0, lower_diag && i == j)});
                        }
                    if (pass == 0) ls.ncrit = v.size() - ls.u0;
                }
                // below the band: column k+1 with all of the panel so far
                if (jc < k1)
                    for (int i = jc + 2; i < T; ++i)
                        v.push_back(TileTask{toff(i, jc, ld), toff(i, k0, ld), toff(jc, k0, ld), kl,
                                             BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0)});
                if (jc < k1) {
                    for (int cc = 0; cc < k0; ++cc)
                        v.push_back(TileTask{toff(jc, cc, ld), toff(jc, k0, ld), toff(k0, cc, ld), kl,
                                             BUF_X, BUF_B, BUF_X, tile_modes(CM_SUB, 0, 1)});
                    for (int cc = k0; cc <= k; ++cc)
                        v.push_back(TileTask{toff(jc, cc, ld), toff(jc, cc, ld), toff(cc, cc, ld), (k + 1 - cc) * GPRN_TILE,
                                             BUF_X, BUF_B, BUF_X, tile_modes(CM_SETNEG, 0, 1)});
                }
                ls.nu = v.size() - ls.u0;
            }
        }
        // ---- "first" and "next" column by column (GPRN_EAGER_NEXT): at step k of the panel, behind its panel products,
        // column k's K = 128 share of the update of the next panel's columns / rows (and of the panel after next's diagonal
        // and sub-diagonal tiles) -- the same tiles, the same additions in the same order as the K = 512 launches at the
        // panel boundary, three quarters of them before the boundary
        if (set == 0) {
            if (c->esteps.size() != (size_t)T) c->esteps.assign(T, gprn_ctx::EStep{0, 0});
            const int n1e = std::min(T, k1 + outer), n2e = std::min(T, n1e + outer);
            for (int k = k0; k < k1; ++k) {
                gprn_ctx::EStep& es = c->esteps[k];
                es.e0 = v.size();
                const uint8_t ft = (k == 0) ? 32 : 0;                      // the first panel's first column: first touch
                for (int j = k1; j < n1e; ++j)
                    for (int i = j + 2; i < T; ++i)
                        v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                             BUF_B, BUF_B, BUF_B, (uint8_t)(tile_modes(CM_SUB, 0, 0) | ft)});
                for (int j = n1e; j < n2e; ++j)
                    for (int i = j; i < std::min(T, j + 2); ++i)
                        v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                             BUF_B, BUF_B, BUF_B, (uint8_t)(tile_modes(CM_SUB, 0, 0, lower_diag && i == j) | ft)});
                for (int i = k1; i < n1e; ++i)
                    for (int cc = 0; cc <= k; ++cc)
                        v.push_back(TileTask{toff(i, cc, ld), toff(i, k, ld), toff(k, cc, ld), GPRN_TILE,
                                             BUF_X, BUF_B, BUF_X, tile_modes(cc == k ? CM_SETNEG : CM_SUB, 0, 1)});
                es.ne = v.size() - es.e0;
            }
        }
        gprn_ctx::OuterRange o{k0, k1, 0, 0, 0, 0, 0, 0, 0, 0, 0, {0}, {0}, 0, 0, 0, 0};
        const int kw = (k1 - k0) * GPRN_TILE;
        const int n1 = std::min(T, k1 + outer);   // the next panel is tiles [k1, n1)
        // pass 0 ("first"): the next panel's first column of B / first row of R -- what stream3's half of
        // its first tile step needs; pass 1 ("next"): the rest of the next panel's columns / rows;
        // pass 2 ("rest"): everything beyond.
        // The next panel's diagonal and sub-diagonal tiles are not touched here at all: the steps of this
        // panel have brought them up to date already (see the in-panel lists).  The same tiles of the panel
        // after next belong to "next" rather than "rest": the next panel's steps start updating them as soon
        // as "next" is done, while "rest" may still be running.
        const int n2 = std::min(T, n1 + outer);
        auto clsB = [&](int i, int j) {
            if (j < n1 && i <= j + 1) return -2;
            if (j == k1) return 0;
            if (j < n1 || (j < n2 && i <= j + 1)) return 1;
            return 2;
        };
        auto clsR = [&](int i) { return i == k1 ? 0 : (i < n1 ? 1 : 2); };
        for (int pass = 0; pass < 3; ++pass) {
            const size_t begin = v.size();
            for (int i = k1; i < T; ++i) {
                for (int j = k1; j <= i; ++j) {
                    if (clsB(i, j) != pass) continue;
                    // (bit 5: the first outer panel's update is the first K = 512 update of every tile it touches)
                    v.push_back(TileTask{toff(i, j, ld), toff(i, k0, ld), toff(j, k0, ld), kw,
                                         BUF_B, BUF_B, BUF_B,
                                         (uint8_t)(tile_modes(CM_SUB, 0, 0, lower_diag && i == j) | (k0 == 0 ? 32 : 0))});
                }
                if (clsR(i) != pass) continue;
                for (int cc = 0; cc < k0; ++cc)
                    v.push_back(TileTask{toff(i, cc, ld), toff(i, k0, ld), toff(k0, cc, ld), kw,
                                         BUF_X, BUF_B, BUF_X, tile_modes(CM_SUB, 0, 1)});
                for (int cc = k0; cc < k1; ++cc)
                    v.push_back(TileTask{toff(i, cc, ld), toff(i, cc, ld), toff(cc, cc, ld),
                                         (k1 - cc) * GPRN_TILE, BUF_X, BUF_B, BUF_X,
                                         tile_modes(CM_SETNEG, 0, 1)});
            }
            if (pass == 0) {
                o.first0 = begin; o.nfirst = v.size() - begin;
                // The same update in two parts (factor_invert_split, GPRN_SPLIT_FIRST): what the panel's columns but the
                // last contribute -- everything it needs exists once the panel's last-but-one step has its panel products,
                // and stream3 has nothing to do while the last diagonal block runs -- and the last column's share (K = 128),
                // all that is left at the panel boundary, where the chain's next step waits for it.  The additions keep their
                // order (k ascending), so the result is the same to the last bit.
                if (k1 - k0 >= 2 && o.nfirst > 0) {
                    const size_t nf = o.nfirst;
                    const int ka = (k1 - 1 - k0) * GPRN_TILE;                 // K of part a
                    o.fa0 = v.size();
                    for (size_t t = 0; t < nf; ++t) {
                        TileTask a = v[o.first0 + t];
                        const int cm = a.modes & 3;
                        if (cm == CM_SUB) a.klen = ka;                      // operands start at the panel's first column
                        else a.klen -= GPRN_TILE;                           // first touch of R_ic: columns c .. k1-2 (none for c = k1-1)
                        if (a.klen > 0) v.push_back(a);
                    }
                    o.nfa = v.size() - o.fa0;
                    o.fb0 = v.size();
                    for (size_t t = 0; t < nf; ++t) {
                        TileTask b = v[o.first0 + t];
                        const int cm = b.modes & 3;
                        const bool touched = cm == CM_SUB || b.klen > GPRN_TILE;     // part a has written the tile
                        const int64_t skip = cm == CM_SUB ? ka : b.klen - GPRN_TILE; // K already done
                        b.a_off += skip;                                              // a_mode 0: k contiguous
                        b.b_off += ((b.modes >> 3) & 1) ? (int64_t)skip * ld : skip;  // b_mode 1: k along rows
                        b.klen = GPRN_TILE;
                        if (touched) b.modes = (uint8_t)((b.modes & ~(3 | 32)) | CM_SUB);
                        v.push_back(b);
                    }
                    o.nfb = v.size() - o.fb0;
                }
            }
            else if (pass == 1) {
                o.next0 = begin; o.nnext = v.size() - begin;
                if (outer <= GPRN_OUTER) {
                    // by the column of B / the row of R inside the next panel (the left-looking steps need them one per step);
                    // the panel after next's diagonal and sub-diagonal tiles go with the first group
                    auto grp = [&](const TileTask& t) {
                        const int i = (int)(t.c_off / ((int64_t)GPRN_TILE * ld)), j = (int)((t.c_off % ld) / GPRN_TILE);
                        const int g = t.c_buf == BUF_B ? (j < n1 ? j - k1 : 1) : i - k1;
                        return std::min(std::max(g, 1), outer - 1);
                    };
                    std::stable_sort(v.begin() + begin, v.end(), [&](const TileTask& a, const TileTask& b) { return grp(a) < grp(b); });
                    size_t at = begin;
                    for (int g = 1; g < outer; ++g) {
                        o.grp0[g] = at;
                        while (at < v.size() && grp(v[at]) == g) ++at;
                        o.ngrp[g] = at - o.grp0[g];
                    }
                }
            }
            else {
                // "rest" in two parts: A = what the NEXT panel's outer update writes again (the columns / rows of the
                // panel after next, and of the one after that its diagonal and sub-diagonal tiles), B = the others.
                // The next panel's "first" and "next" launches wait for A only (F_RESTA).
                const int n3 = std::min(T, n2 + outer);
                auto in_a = [&](const TileTask& t) {
                    const int i = (int)(t.c_off / ((int64_t)GPRN_TILE * ld)), j = (int)((t.c_off % ld) / GPRN_TILE);
                    return t.c_buf == BUF_B ? (j < n2 || (j < n3 && i <= j + 1)) : i < n2;
                };
                std::stable_partition(v.begin() + begin, v.end(), in_a);
                size_t na = 0;
                while (begin + na < v.size() && in_a(v[begin + na])) ++na;
                // Workgroups are dispatched in task order and are not preempted: with the short
                // first-touch tasks (K = 128..384) in front, the first slots free up after a
                // quarter of a full task instead of all at once.
                auto by_klen = [](const TileTask& a, const TileTask& b) { return a.klen < b.klen; };
                std::stable_sort(v.begin() + begin, v.begin() + begin + na, by_klen);
                std::stable_sort(v.begin() + begin + na, v.end(), by_klen);
                o.rest0 = begin; o.nrest = v.size() - begin; o.nrestA = na;
            }
        }
        outers.push_back(o);
    }
    }   // set
    // ---- block schedule (factor_invert_blocks).  Outer panel [k0, k1) of GPRN_OUTER tiles; D = its diagonal block.
    //   per tile step k: in-block panel   L_ik = B_ik X_kk^T (k < i < k1),   X_kc = X_kk R_kc (k0 <= c < k)
    //                    in-block update  B_ij -= L_ik L_jk^T (k < j <= i < k1),   R_ic (-)= L_ik X_kc (k < i < k1, k0 <= c <= k)
    //   -> L_D and X_D = L_D^-1 (the diagonal block of X IS the inverse of the diagonal block of L)
    //   once per panel:  L[i, panel] = B[i, panel] X_D^T (i >= k1),   X[panel, c] = X_D R[panel, c] (c < k0)
    // as plain products with K = 128 (j' + 1) for column / row tile j' of the panel (X_D is lower triangular).  Both
    // would overwrite their own inputs tile by tile, so they go to MIRRORS, transposed, in the unused strictly upper
    // tiles of the OTHER buffer's ... of a buffer:  L[i, k0+j']^T -> BUF_X tile (k0+j', i),  X[k0+j', c]^T -> BUF_B tile
    // (c, k0+j').  Transposed, the four tiles of a panel row are contiguous in k for the trailing update, which reads
    // its operands from the mirrors (modes 1/1 for L L^T, 1/0 for L X).  X's rows are copied back into place (the
    // phase's reductions read them); L's only for callers that ask (fast_factor = false).
    {
        const int outer = outer_big;
        c->bsteps.assign(T, gprn_ctx::BlkStep{0, 0, 0, 0, 0});
        c->bpanels.clear();
        for (int k0 = 0; k0 < T; k0 += outer) {
            const int k1 = std::min(T, k0 + outer), n1 = std::min(T, k1 + outer), n2 = std::min(T, n1 + outer);
            for (int k = k0; k < k1; ++k) {
                gprn_ctx::BlkStep& b = c->bsteps[k];
                b.l0 = v.size();
                for (int i = k + 1; i < k1; ++i)
                    v.push_back(TileTask{toff(i, k, ld), toff(i, k, ld), toff(k, k, ld), GPRN_TILE,
                                         BUF_B, BUF_B, BUF_X, tile_modes(CM_SET, 0, 0)});
                b.nl_l = v.size() - b.l0;
                for (int cc = k0; cc < k; ++cc)
                    v.push_back(TileTask{toff(k, cc, ld), toff(k, k, ld), toff(k, cc, ld), GPRN_TILE,
                                         BUF_X, BUF_X, BUF_X, tile_modes(CM_SET, 0, 1)});
                b.nl = v.size() - b.l0;
                b.u0 = v.size();
                for (int j = k + 1; j < k1; ++j)           // the next diagonal tile first
                    for (int i = j; i < k1; ++i)
                        v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                             BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0, lower_diag && i == j)});
                for (int i = k + 1; i < k1; ++i)
                    for (int cc = k0; cc <= k; ++cc)
                        v.push_back(TileTask{toff(i, cc, ld), toff(i, k, ld), toff(k, cc, ld), GPRN_TILE,
                                             BUF_X, BUF_B, BUF_X, tile_modes(cc == k ? CM_SETNEG : CM_SUB, 0, 1)});
                b.nu = v.size() - b.u0;
            }
            gprn_ctx::BlkPanel bp{k0, k1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int kw = (k1 - k0) * GPRN_TILE;
            // L side: C = X_D[j', 0..j'] B[i, k0..k0+j']^T, tile (k0+j', i) of BUF_X
            bp.tl0 = v.size();
            for (int i = k1; i < T; ++i) {
                for (int jp = 0; jp < k1 - k0; ++jp)
                    v.push_back(TileTask{toff(k0 + jp, i, ld), toff(k0 + jp, k0, ld), toff(i, k0, ld), (jp + 1) * GPRN_TILE,
                                         BUF_X, BUF_X, BUF_B, tile_modes(CM_SET, 0, 0)});
                if (i == n1 - 1) bp.ntl_early = v.size() - bp.tl0;
            }
            bp.ntl = v.size() - bp.tl0;
            // X side: C = R[panel, c]^T X_D[j', 0..j']^T, tile (c, k0+j') of BUF_B
            bp.tx0 = v.size();
            for (int cc = 0; cc < k0; ++cc)
                for (int jp = 0; jp < k1 - k0; ++jp)
                    v.push_back(TileTask{toff(cc, k0 + jp, ld), toff(k0, cc, ld), toff(k0 + jp, k0, ld), (jp + 1) * GPRN_TILE,
                                         BUF_B, BUF_X, BUF_X, tile_modes(CM_SET, 1, 0)});
            bp.ntx = v.size() - bp.tx0;
            // copies (k_tile_tcopy: C = A^T): X rows back into place; L far tiles into place (on request)
            bp.cbx0 = v.size();
            for (int cc = 0; cc < k0; ++cc)
                for (int jp = 0; jp < k1 - k0; ++jp)
                    v.push_back(TileTask{toff(k0 + jp, cc, ld), toff(cc, k0 + jp, ld), 0, 0, BUF_X, BUF_B, BUF_B, 0});
            bp.cbl0 = v.size();
            for (int i = k1; i < T; ++i)
                for (int jp = 0; jp < k1 - k0; ++jp)
                    v.push_back(TileTask{toff(i, k0 + jp, ld), toff(k0 + jp, i, ld), 0, 0, BUF_B, BUF_X, BUF_X, 1 /* clear the mirror */});
            // trailing update from the mirrors.  Classes: 0 the next panel's diagonal block (the chain's own launch),
            // 1 the rest of the next panel's columns of B and its rows of R, 2 the panel after next's, 3 beyond.
            // (the DIAGONAL blocks of the panels after that move up one class each -- the one after next with "next",
            // the third with "ahead": the chain's update of a diagonal block then follows launches that ran a whole
            // panel earlier, not the previous panel's "ahead" part, which queues behind a long "bulk" launch)
            const int n3 = std::min(T, n2 + outer);
            auto clsB = [&](int i, int j) {
                if (j < n1) return i < n1 ? 0 : 1;
                if (j < n2) return i < n2 ? 1 : 2;
                if (j < n3) return i < n3 ? 2 : 3;
                return 3;
            };
            auto clsR = [&](int i) { return i < n1 ? 1 : (i < n2 ? 2 : 3); };
            for (int pass = 0; pass < 4; ++pass) {
                const size_t begin = v.size();
                for (int i = k1; i < T; ++i) {
                    for (int j = k1; j <= i; ++j) {
                        if (clsB(i, j) != pass) continue;
                        v.push_back(TileTask{toff(i, j, ld), toff(k0, i, ld), toff(k0, j, ld), kw,
                                             BUF_B, BUF_X, BUF_X, tile_modes(CM_SUB, 1, 1, lower_diag && i == j)});
                    }
                    if (clsR(i) != pass) continue;
                    for (int cc = 0; cc < k0; ++cc)
                        v.push_back(TileTask{toff(i, cc, ld), toff(k0, i, ld), toff(cc, k0, ld), kw,
                                             BUF_X, BUF_X, BUF_B, tile_modes(CM_SUB, 1, 0)});
                    for (int cc = k0; cc < k1; ++cc)
                        v.push_back(TileTask{toff(i, cc, ld), toff(cc, i, ld), toff(cc, cc, ld), (k1 - cc) * GPRN_TILE,
                                             BUF_X, BUF_X, BUF_X, tile_modes(CM_SETNEG, 1, 1)});
                }
                auto by_klen = [](const TileTask& a, const TileTask& b) { return a.klen < b.klen; };
                if (pass == 0) { bp.dn0 = begin; bp.ndn = v.size() - begin; }
                else if (pass == 1) { bp.next0 = begin; bp.nnext = v.size() - begin; }
                else if (pass == 2) {
                    std::stable_sort(v.begin() + begin, v.end(), by_klen);
                    bp.rest0 = begin; bp.nrestA = v.size() - begin;
                } else {
                    std::stable_sort(v.begin() + begin, v.end(), by_klen);
                    bp.nrest = v.size() - bp.rest0;
                }
            }
            c->bpanels.push_back(bp);
        }
    }
    // lower(X^T X) -> BUF_B: tile (a,b), a >= b, sums over rows a*128 .. ld of X
    // (short contractions first: a launch of these runs beside the next phase's factorisation, whose diagonal
    // block needs a whole free CU -- the CUs that got the short tasks come free within tens of microseconds)
    c->lauum0 = v.size();
    for (int a = T - 1; a >= 0; --a)
        for (int b = 0; b <= a; ++b)
            v.push_back(TileTask{toff(a, b, ld), toff(a, a, ld), toff(a, b, ld), ld - a * GPRN_TILE,
                                 BUF_B, BUF_X, BUF_X, tile_modes(CM_SET, 1, 1)});
    c->nlauum = v.size() - c->lauum0;

    if (v.size() > c->tasks_cap) {
        if (c->d_tasks) hipFree(c->d_tasks);
        c->d_tasks = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_tasks, v.size() * sizeof(TileTask)));
        c->tasks_cap = v.size();
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream2));
    HIP_TRY(c, hipMemcpyAsync(c->d_tasks, v.data(), v.size() * sizeof(TileTask),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tasks_T = T;
    return GPRN_OK;
}

// One thread on a stream: raise a flag for whoever waits on the work before it, then hold the stream
// until another flag is up -- a stream write and a stream wait of the runtime (two 5 us kernels) in one.
__global__ void k_flag_sync(unsigned* raise_flag, unsigned raise_value, const unsigned* wait_flag,
                            unsigned wait_value, unsigned* timed_out)
{
    if (threadIdx.x != 0) return;
    if (raise_flag) __hip_atomic_store(raise_flag, raise_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (wait_flag) spin_until(wait_flag, wait_value, timed_out);
}

// ... and several of each in one: every stream memory operation of this runtime is a 4-5 us kernel of its own, and at a
// panel boundary stream3 had five of them in a row, the chain stream four at the end of a factorisation
struct FlagOps { unsigned* raise[2]; const unsigned* wait[4]; };
__global__ void k_flag_multi(FlagOps ops, unsigned value, unsigned* timed_out)
{
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (ops.raise[i]) __hip_atomic_store(ops.raise[i], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ops.wait[i]) spin_until(ops.wait[i], value, timed_out);
}

#define GPRN_FLAG_KINDS 11          // flag kinds per tile step / outer panel (factor_invert_split)

// Flags or events for this context?  Kernels that wait for other kernels need those to be able to run
// beside them: every switch that serialises kernels or starves the hardware queues means events.
//   rocprofv3 --pmc (ROCPROF_COUNTER_COLLECTION=1), AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING,
//   GPU_MAX_HW_QUEUES < 4 (three streams of this context + the null stream), no stream memory operations.
// GPRN_FLAGS=0/1 overrides; gprn_set_option(ctx, "flags", v) sets it per context; a time-out latches 0.
__global__ void k_flag_sync(unsigned* raise_flag, unsigned raise_value, const unsigned* wait_flag,
                            unsigned wait_value, unsigned* timed_out);

// Do kernels of the chain stream and of stream4 run side by side?  With too few hardware queues the runtime
// folds two streams onto one, and a kernel that waits in-kernel for a later launch of the "other" stream would
// never see it start.
// true when a kernel on `waiter` that polls (20 ms at most) for a flag sees a kernel launched AFTER it on
// `producer` raise it -- i.e. the two streams sit on different hardware queues
static bool streams_overlap(hipStream_t waiter, hipStream_t producer)
{
    unsigned* w = nullptr;                         // [0] flag, [2] time-out word, [3] budget in 100 MHz ticks, [4] which flag
    if (hipMalloc(&w, 8 * sizeof(unsigned)) != hipSuccess) return false;
    const unsigned init[8] = {0u, 0u, 0u, 2000000u, 0u, 0u, 0u, 0u};
    bool ok = hipMemcpy(w, init, sizeof(init), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, waiter, (unsigned*)nullptr, 0u, (const unsigned*)w, 1u, w + 2);
        hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, producer, w, 1u, (const unsigned*)nullptr, 0u, w + 2);
        ok = hipStreamSynchronize(waiter) == hipSuccess && hipStreamSynchronize(producer) == hipSuccess;
    }
    unsigned out[8] = {0, 0, 1, 0, 0, 0, 0, 0};
    if (ok) ok = hipMemcpy(out, w, sizeof(out), hipMemcpyDeviceToHost) == hipSuccess;
    hipFree(w);
    return ok && out[0] == 1u && out[2] == 0u;
}

int factor_probe_streams(gprn_ctx* c)
{
    if (c->chain_streams >= 0) return c->chain_streams;
    c->chain_streams = (c->stream4 && streams_overlap(c->stream4, c->stream) && streams_overlap(c->stream, c->stream4)) ? 1 : 0;
    return c->chain_streams;
}

// Device-side flags or HIP events?  Flags need (1) stream memory operations, (2) no tool or setting that runs one
// kernel at a time (counter collection, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, fewer than 4 hardware queues),
// (3) the chain, side and bulk streams on different hardware queues -- probed once per device (the streams are
// shared by every context on it): six tiny launch pairs, each a kernel polling for a flag raised by a kernel
// launched after it on another stream.  GPRN_FLAGS=0/1 overrides (1) still needs the stream memory operations.
int factor_use_flags(gprn_ctx* c)
{
    if (c->use_flags >= 0) return c->use_flags;
    auto on = [](const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; };
    const char* e = getenv("GPRN_FLAGS");
    const char* hq = getenv("GPU_MAX_HW_QUEUES");
    const bool serialised = on("ROCPROF_COUNTER_COLLECTION") || on("AMD_SERIALIZE_KERNEL") ||
                            on("HIP_LAUNCH_BLOCKING") || (hq && atoi(hq) > 0 && atoi(hq) < 4);
    int can = 0;                               // stream memory operations are optional in HIP
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess) can = 0;
    if (e) return c->use_flags = (atoi(e) != 0 && can) ? 1 : 0;
    if (serialised || !can) return c->use_flags = 0;
    DeviceStreams* d = c->shared;
    if (d && d->use_flags < 0) {
        hipStream_t st[3] = {c->stream, c->stream2, c->stream3};
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i)
            for (int j = 0; j < 3 && ok; ++j)
                if (i != j) ok = streams_overlap(st[i], st[j]);
        d->use_flags = ok ? 1 : 0;
        if (!ok)
            fprintf(stderr, "[gprn] device %d: the library's streams share a hardware queue; cross-stream dependencies "
                            "go through HIP events\n", c->device);
    }
    return c->use_flags = d ? d->use_flags : 1;
}

// Split schedule.  Per tile step k the only launches on the chain stream are the diagonal
// block, the ONE panel tile below it (L_{k+1,k}) and the ONE in-panel update that completes the
// next diagonal tile (B_{k+1,k+1}); the remaining panel tiles and in-panel updates of the step run on
// `stream3` beside the next diagonal block.  Order of read-modify-writes on a tile is kept by
// events: stream3 starts step k's panel after diag(k) and its updates after L_{k+1,k}; the chain
// takes L_{k+2,k+1} only after stream3 finished step k's updates.  The last step of an outer
// panel stays whole on the chain (the outer update needs all of it).
static int factor_invert_split(gprn_ctx* c, int nbatch, int set)
{
    int rc;
    hipStream_t s0 = c->stream, s1 = c->stream3, s2 = c->stream2;
    static size_t big = 0;                         // tasks x batch above which 128x128 workgroups pay
    if (!big) { const char* e = getenv("GPRN_FEW_TASKS"); big = e && atoi(e) > 0 ? (size_t)atoi(e) : 4000; }
    auto shape_upd = [&](size_t n) { return n * (size_t)nbatch > big ? TS_128x128 : TS_64x64; };
    // In this schedule the 64x128 and 128x64 shapes are used by the panel products only, whose B resp. A
    // operand is the triangular X_kk: their kernels skip the block products that only meet its zero half
    // (+2 % sweeps/s where the chain dominates; applying it to the chain's launch alone measured -2 %,
    // a second code object for one small launch per step).  GPRN_TRI=0 switches it off.
    static int tri = -1;
    if (tri < 0) { const char* e = getenv("GPRN_TRI"); tri = e ? atoi(e) : 1; }
    // tag: TG_PANEL for the panel products (the only users of the 64x128 / 128x64 shapes), else as given
    auto tiles = [&](size_t first, size_t n, hipStream_t st, int shape, int fam = GPRN_T_PANEL,
                     Signal sig = Signal{nullptr, 0}, Await aw = Await{nullptr, 0, nullptr}, int tag = TG_INNER) {
        if ((shape == TS_64x128 || shape == TS_128x64) && tag == TG_INNER) {   // (the panel products' calls leave the tag alone)
            tag = TG_PANEL;
            if (tri) shape = shape == TS_64x128 ? TS_64x128_BTRI : TS_128x64_ATRI;
        }
        return launch_tiles(c, c->d_tasks + first, n, c->d_ptrs, nbatch, c->ld, fam, st, shape, sig, aw, tag);
    };
    // Cross-stream dependencies travel through 32-bit flags in device memory instead of events:
    // hipStreamWriteValue32 / hipStreamWaitValue32 cost less than an event record / wait pair
    // (+3 % sweeps/s at N = 4096, +14 % at N = 2048), and the chain's two small kernels raise
    // their flag themselves (Signal), so nothing at all sits between the chain's three dependent
    // launches.  Flags only grow: a call waits for its own epoch.  GPRN_FLAGS=0: events.
    const int use_flags = factor_use_flags(c);
    auto side_stamp = [&](int k, int i) {          // GPRN_STEP_STAMPS=2: the clock on stream3 at this point of step k
        if (c->side_stamps) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s1, c->side_stamps + (size_t)k * 8 + i);
    };
    enum { F_DIAG = 0, F_MINIL, F_INNER, F_PANEL, F_NEXT, F_REST, F_FIRST, F_XW, F_U, F_RESTA, F_TAIL, F_KINDS };
    static_assert(F_KINDS == GPRN_FLAG_KINDS, "factor_check_waits reads the word behind T * GPRN_FLAG_KINDS flag pairs");
    if (use_flags && c->sig_T < c->T) {
        if (c->d_sig) hipFree(c->d_sig);
        c->d_sig = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_sig, ((size_t)c->T * F_KINDS * 2 + 4) * sizeof(unsigned)));
        HIP_TRY(c, hipMemset(c->d_sig, 0, ((size_t)c->T * F_KINDS * 2 + 4) * sizeof(unsigned)));
        c->sig_T = c->T;
        c->epoch = 0;
        c->sig_budget_ms = -1;
    }
    if (use_flags && c->sig_budget_ms != c->wait_budget_ms) {
        // the word behind the sticky time-out word: budget of one in-kernel wait, 100 MHz ticks
        const unsigned ticks = (unsigned)std::min<long long>(0xffffffffll, (long long)c->wait_budget_ms * 100000ll);
        HIP_TRY(c, hipMemcpy(c->d_sig + (size_t)c->sig_T * F_KINDS * 2 + 1, &ticks, sizeof(unsigned), hipMemcpyHostToDevice));
        c->sig_budget_ms = c->wait_budget_ms;
    }
    const unsigned epoch = ++c->epoch;
    hipEvent_t events[F_KINDS] = {c->ev_diag, c->ev_minil, c->ev_inner, c->ev_panel, c->ev_next, c->ev_rest, c->ev_first, nullptr, nullptr, c->ev_resta, c->ev_tail};
    auto slot = [&](int idx, int kind) { return c->d_sig + ((size_t)idx * F_KINDS + kind) * 2; };
    auto in_kernel = [&](int idx, int kind) {      // the launch raises the flag itself
        return use_flags ? Signal{slot(idx, kind), epoch} : Signal{nullptr, 0};
    };
    int inner_raises = 0;
    auto withheld = [&](int kind) {                // test hook (gprn_set_option "withhold_inner")
        return use_flags && kind == F_INNER && c->withhold_inner > 0 && ++inner_raises == c->withhold_inner;
    };
    auto raise = [&](hipStream_t st, int idx, int kind) {
        if (withheld(kind)) return hipSuccess;
        return use_flags ? hipStreamWriteValue32(st, slot(idx, kind) + 1, epoch, 0) : hipEventRecord(events[kind], st);
    };
    // the chain's L_{k+1,k} launch (2 workgroups per matrix) waits for stream3's flag itself: the
    // stream wait is a 5 us kernel of its own on this runtime (__amd_rocclr_streamOpsWait)
    auto in_kernel_wait = [&](int idx, int kind) {
        return Await{slot(idx, kind) + 1, epoch, c->d_sig + (size_t)c->sig_T * F_KINDS * 2, nullptr, 0};
    };
    auto await = [&](hipStream_t st, int idx, int kind) {
        return use_flags ? hipStreamWaitValue32(st, slot(idx, kind) + 1, epoch, hipStreamWaitValueGte, 0xffffffffu)
                         : hipStreamWaitEvent(st, events[kind], 0);
    };
    int rest_J = -1, next_J = -1, first_J = -1;    // outer panels whose rest / next / first update is not joined yet
    int inner_k = -1;                              // tile step whose F_INNER flag stream3 still has to raise
    unsigned* timed_out = c->d_sig ? c->d_sig + (size_t)c->sig_T * F_KINDS * 2 : nullptr;
    // stream3 at the start of step k: raise F_INNER of the step before, then wait for diag(k)
    auto side_sync = [&](int k) -> int {
        if (!use_flags) {
            if (inner_k >= 0) HIP_TRY(c, raise(s1, inner_k, F_INNER));
            inner_k = -1;
            HIP_TRY(c, await(s1, k, F_DIAG));
            return GPRN_OK;
        }
        const bool skip = inner_k >= 0 && withheld(F_INNER);
        hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, s1,
                           inner_k >= 0 && !skip ? slot(inner_k, F_INNER) + 1 : (unsigned*)nullptr, epoch,
                           (const unsigned*)(slot(k, F_DIAG) + 1), epoch, timed_out);
        inner_k = -1;
        HIP_TRY(c, hipGetLastError());
        return GPRN_OK;
    };
    // (never withheld by the test hook: this raise is also what the last tile step's STREAM wait consumes, and a
    // stream wait has no time-out -- the hook only ever drops raises whose consumers wait in-kernel)
    auto flush_inner = [&]() -> int {              // nothing else follows on stream3 soon
        if (inner_k >= 0)
            HIP_TRY(c, use_flags ? hipStreamWriteValue32(s1, slot(inner_k, F_INNER) + 1, epoch, 0)
                                 : hipEventRecord(events[F_INNER], s1));
        inner_k = -1;
        return GPRN_OK;
    };
    // the X part of the panel is a small launch: its last workgroup holds it open until L_{k+1,k} is
    // there, so the in-panel updates behind it need no stream wait
    bool eager = false;                            // (set below: GPRN_EAGER_NEXT)
    auto x_part_then = [&](int k) {
        // (eager: F_XW goes up when the panel products are in memory -- the next-panel stream waits for it)
        return use_flags ? Signal{slot(k, F_XW), eager ? epoch : 0u, slot(k, F_MINIL) + 1, epoch, timed_out}
                         : Signal{nullptr, 0, nullptr, 0, nullptr};
    };
    const Await noaw{nullptr, 0, nullptr};
    const Signal nosig{nullptr, 0, nullptr, 0, nullptr};
    // GPRN_CHAIN=1: the chain as one persistent workgroup per matrix (k_chain; flag schedule only, it waits
    // in-kernel) instead of three launches per tile step on the chain stream.  Off by default: one CU does
    // L_{k+1,k} and the B_{k+1,k+1} update in 12 + 22 us where the launches spread them over six CUs, and the
    // step is no shorter (measured 86 vs 97 sweeps/s at config 3, 365 vs 423 at config 2).
    static int chain_env = -1;
    if (chain_env < 0) { const char* e = getenv("GPRN_CHAIN"); chain_env = e ? atoi(e) : 0; }
    const bool use_chain = use_flags && chain_env == 1 && c->T > 1;
    static int stamps_env = -1;                    // GPRN_CHAIN_STAMPS=1: clock stamps of matrix 0's chain (probes)
    if (stamps_env < 0) { const char* e = getenv("GPRN_CHAIN_STAMPS"); stamps_env = e ? atoi(e) : 0; }
    if (use_chain && stamps_env && c->stamps_T < c->T) {
        if (c->d_stamps) hipFree(c->d_stamps);
        HIP_TRY(c, hipMalloc(&c->d_stamps, (size_t)c->T * 8 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_stamps, 0, (size_t)c->T * 8 * sizeof(unsigned long long)));
        c->stamps_T = c->T;
    }
    if (use_chain) {
        const int outer_w = c->outers[set][0].k1 - c->outers[set][0].k0;
        prof_begin(c, GPRN_T_DIAG, s0);
        hipLaunchKernelGGL(k_chain, dim3(nbatch), dim3(256), 0, s0, (double* const*)c->d_ptrs, c->ld, c->T,
                           outer_w, c->d_info_cur, c->d_sig, (int)F_KINDS, (int)F_DIAG, (int)F_MINIL, (int)F_INNER,
                           (int)F_FIRST, epoch, timed_out, c->d_stamps);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());
        if (c->chain_started) {
            // work that must not take the chain's CUs before it is resident (run_phase: the X^T X product of
            // the node phase, which fills the head of the weight phase): behind the first diagonal block
            HIP_TRY(c, await(s2, 0, F_DIAG));
            std::function<int()> f;
            f.swap(c->chain_started);
            if ((rc = f())) return rc;
        }
    }
    // GPRN_CHAIN_STREAMS=1: the chain on two streams when the runtime runs them side by side (measured: no gain,
    // 96.7 vs 97.3 sweeps/s at config 3 -- what a chain kernel costs beyond its arithmetic is inside it, fences
    // and operand latency, not its dispatch); default: one stream
    static int cs_env = -1;
    if (cs_env < 0) { const char* e = getenv("GPRN_CHAIN_STREAMS"); cs_env = e ? atoi(e) : 0; }
    static int persist_max = -1;                   // GPRN_CHAIN2_MAX_BATCH (experiments): only for batches up to that
    if (persist_max < 0) { const char* e = getenv("GPRN_CHAIN2_MAX_BATCH"); persist_max = e ? atoi(e) : 1 << 30; }
    const bool persist = use_flags && chain_env == 2 && c->T > 1 && nbatch <= persist_max && factor_probe_streams(c) == 1;
    const bool two_streams = use_flags && !use_chain && (cs_env || persist) && factor_probe_streams(c) == 1;
    if (persist) {
        prof_begin(c, GPRN_T_DIAG, s0);
        PtrArgs pa;
    pa.stamps = nullptr;
        // GPRN_DIAG_EXCL_KB (experiments): unused dynamic LDS on top of the kernel's 46.6 KB -- with enough of it no
        // other workgroup that uses LDS shares the persistent workgroup's CU (no co-resident MFMA waves on its SIMDs)
        static int excl_kb = -1;
        if (excl_kb < 0) { const char* e = getenv("GPRN_DIAG_EXCL_KB"); excl_kb = e ? atoi(e) : 0; }
        const size_t dyn_excl = std::min<size_t>((size_t)excl_kb * 1024, lds_limit(c->device) - DIAG_LDS_DOUBLES * sizeof(double));
        if (tab_rows(c, c->d_ptrs, nbatch, &pa))
            hipLaunchKernelGGL(k_diag_chain<true>, dim3(nbatch), dim3(256), dyn_excl, s0, (double* const*)c->d_ptrs, pa, c->ld, c->T,
                               c->d_info_cur, c->d_sig, (int)F_KINDS, (int)F_DIAG, (int)F_U, epoch, timed_out);
        else
            hipLaunchKernelGGL(k_diag_chain<false>, dim3(nbatch), dim3(256), dyn_excl, s0, (double* const*)c->d_ptrs, pa, c->ld, c->T,
                               c->d_info_cur, c->d_sig, (int)F_KINDS, (int)F_DIAG, (int)F_U, epoch, timed_out);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());
    }
    if (!use_chain && !use_flags && c->chain_started) {    // event schedule: nothing to gate it on
        std::function<int()> f;
        f.swap(c->chain_started);
        if ((rc = f())) return rc;
    }
    bool tail_on_s2 = false;                       // rows_final ran on the bulk stream: joined at the end
    // B is still to be built (run_phase): only what the first outer panel's tile steps touch; its K = 512 update forms
    // the other tiles from K on the way in (bit 5 of their tasks; tile_mma ft_K) -- 16 N^2 bytes of HBM traffic per
    // matrix and three quarters of k_build_B's time at the head of the phase less
    bool ft_fused = false;
    if (c->build_pending) {
        const int pend = c->build_pending;
        c->build_pending = 0;
        const int outer = c->outers[set][0].k1 - c->outers[set][0].k0;
        // (the 64 x 64 tile kernel only: every launch of the first panel's update must use that shape)
        const gprn_ctx::OuterRange& o0 = c->outers[set][0];
        const char* bs = getenv("GPRN_BULK_SHAPE");
        const char* bsb = getenv("GPRN_BULK_SHAPE_BIG");
        const bool small_shapes = shape_upd(o0.nfirst) == TS_64x64 && shape_upd(o0.nnext) == TS_64x64 &&
                                  (!bs || atoi(bs) == TS_64x64) && (!bsb || atoi(bsb) == TS_64x64);
        ft_fused = c->ft_s_phase && pend == nbatch && !use_chain && c->T > outer && small_shapes;
        if ((rc = vec_build_B(c, pend, s0, ft_fused ? 1 : 0, outer))) return rc;
    }
    // GPRN_LEFT=1 (opt-in): the steps' updates in left-looking form below the diagonal band and the "next" part of an outer
    // update in groups, one per column / row of the next panel (ensure_tasks); throughput set, default chain kernels.
    // Bit-identical results (the same additions in the same order); measured equal to the right-looking form: 110.3-110.6
    // vs 110.4-111.0 sweeps/s at config 3, 59.8-60.6 vs 59.3-60.4 at config 4 (with the band left-looking too: 107.6)
    static int left_env = -1;
    if (left_env < 0) { const char* e = getenv("GPRN_LEFT"); left_env = e ? atoi(e) : 0; }
    const bool left = left_env && set == 0 && !use_chain && !persist && c->lsteps.size() == (size_t)c->T;
    std::vector<char> grp_pending(c->T + 1, 0);    // [tile column]: its group of "next" is still to be waited for by stream3
    int last_grp = -1;
    static int split_rest = -1;
    if (split_rest < 0) { const char* e = getenv("GPRN_SPLIT_REST"); split_rest = e ? atoi(e) : 1; }
    const bool sr_all = split_rest && c->stream4 && !two_streams && (split_rest >= 2 || c->T <= 64);
    // GPRN_SPLIT_FIRST=1 (opt-in): the "first" part of a panel's outer update in two launches (ensure_tasks fa0 / fb0): the
    // share of the panel's columns but the last goes out behind the last-but-one step's updates, while the last diagonal
    // block of the panel runs and stream3 would idle; at the boundary only the last column's share (K = 128) is left.
    // Bit-identical; slower: 108.6 vs 110.5 sweeps/s at config 3, 58.2 vs 59.3 at config 4 -- the in-kernel stamps
    // (GPRN_STEP_STAMPS) show the chain's wait at a panel's second step unchanged (35-65 us with two matrices) and a new
    // one at its first: what the chain waits for there is not the "first" launch
    static int split_first = -1;
    if (split_first < 0) { const char* e = getenv("GPRN_SPLIT_FIRST"); split_first = e ? atoi(e) : 0; }
    int first_a_done = -1;
    int pending_outer = -1;                        // outer panel whose trailing update is not enqueued yet
    unsigned* pending_up = nullptr;                // a flag the next folded panel launch on stream3 raises at its start
    // GPRN_PANEL_SYNC=n (default 2): with up to n matrices stream3's synchronisation kernel is folded into the panel
    // launch -- F_INNER of the step before goes up when its first workgroup runs, every workgroup waits for the
    // diagonal block itself: one launch less per step on stream3 (config 2 737 -> 751 sweeps/s, config 3 +0.5 %;
    // with six matrices, whose panel launches are hundreds of workgroups that would all poll: -1 %)
    static int panel_sync = -1, merge_panel = -1;
    if (panel_sync < 0) { const char* e = getenv("GPRN_PANEL_SYNC"); panel_sync = e ? atoi(e) : 2; }
    // GPRN_MERGE_PANEL=0: the two halves of the panel as two launches (the form of round 1)
    if (merge_panel < 0) { const char* e = getenv("GPRN_MERGE_PANEL"); merge_panel = e ? atoi(e) : 1; }
    // GPRN_EAGER_NEXT=n (experiments; default 0): with up to n matrices the next panel's share of an outer update is applied
    // column by column on the next-panel stream (ensure_tasks: esteps) instead of as "first" + "next" at the boundary
    static int eager_next = -1;
    if (eager_next < 0) { const char* e = getenv("GPRN_EAGER_NEXT"); eager_next = e ? atoi(e) : 0; }
    eager = use_flags && set == 0 && !left && !use_chain && !persist && !split_first && sr_all && merge_panel && tri &&
            nbatch <= eager_next && c->esteps.size() == (size_t)c->T;
    // A launch whose EVERY workgroup polls at its head must not be able to fill the device: workgroups are never
    // preempted, so once pollers hold every slot a producer that is not resident yet never becomes so and the flag
    // never rises (round 3, config 5's shape: the last tile step's 2 (T - 1) x 15 = 3810 workgroups polled for an
    // update of stream3 that was still queued behind its panel launch -- gpurun_out/cfg5.err).  Up to half
    // a workgroup per CU they leave room on every CU whatever else they are; above that a one-wave kernel waits.
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || n_cu <= 0) n_cu = 64;
    static int poll_any = -1;                      // GPRN_POLL_ANY=1 (diagnosis of the round-3 time-out only): no bound
    if (poll_any < 0) { const char* e = getenv("GPRN_POLL_ANY"); poll_any = e ? atoi(e) : 0; }
    auto may_poll = [&](size_t nwg) { return use_flags && (poll_any || nwg * 2 <= (size_t)n_cu); };
    auto folds_sync = [&](int k) {                 // tile step k's panel launch takes stream3's synchronisation along
        if (k < 0 || k >= c->T || use_chain) return false;
        const gprn_ctx::StepRange& sk = c->steps[set][k];
        return merge_panel && tri && use_flags && nbatch <= panel_sync && sk.npanel_l > 0 && sk.npanel > 1 &&
               may_poll(2 * (sk.npanel - 1) * (size_t)nbatch);
    };
    auto do_outer = [&](int Jp) -> int {
        const size_t J = (size_t)Jp;
        const gprn_ctx::OuterRange& o = c->outers[set][J];
        pending_outer = -1;
        // Outer update of panel J (K = its width).  stream3, which has seen every tile of the panel: what
        // the chain touches first in the next panel (its first column of B and first row of R, its diagonal
        // and sub-diagonal tiles); bulk stream: the rest of the next panel, then everything beyond.  The
        // chain itself goes straight on with the next diagonal block.
        // GPRN_MULTI_FLAG=0: every raise / wait below as a stream memory operation of its own (before round 3's last session)
        static int multi_flag = -1;
        if (multi_flag < 0) { const char* e = getenv("GPRN_MULTI_FLAG"); multi_flag = e ? atoi(e) : 1; }
        const bool multi = use_flags && multi_flag;
        if (!multi) HIP_TRY(c, raise(s1, (int)J, F_PANEL));
        // GPRN_SPLIT_REST=1 (default): the previous panel's "rest" went out as two launches and only the first (A: the
        // tiles this panel's outer update writes again) is waited for here; "next" runs on a stream of its own instead
        // of queueing behind the previous panel's whole "rest" on the bulk stream.  0: one launch, one stream.
        // (measured +2.1 % sweeps/s at N = 4096 and 8192; at N = 16384, where a "rest" launch runs for 11 ms, -0.7 %: up to
        // 64 tile steps by default, GPRN_SPLIT_REST=2 forces it)
        const bool sr = sr_all;
        hipStream_t sn = sr ? c->stream4 : s2;
        const bool eager_J = eager && o.k1 < c->T;     // this panel's "first" + "next" went out column by column
        if (multi) {
            // F_PANEL up, then the two waits, in ONE kernel on stream3 (three operations, 13 us, before)
            FlagOps ops = {{slot((int)J, F_PANEL) + 1, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
            if (rest_J >= 0) ops.wait[0] = slot(rest_J, sr ? F_RESTA : F_REST) + 1;
            if (next_J >= 0) { ops.wait[1] = slot(next_J, F_NEXT) + 1; next_J = -1; }
            if (eager_J) ops.wait[2] = slot((int)J, F_NEXT) + 1;          // (raised behind the panel's last column's share)
            hipLaunchKernelGGL(k_flag_multi, dim3(1), dim3(64), 0, s1, ops, epoch, timed_out);
            HIP_TRY(c, hipGetLastError());
        } else {
        if (rest_J >= 0) HIP_TRY(c, await(s1, rest_J, sr ? F_RESTA : F_REST));      // same tiles as the previous panel's rest / next
        if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
        }
        // the first panel's update forms the tiles of B it touches from K (run_phase built only the others)
        struct FtScope { gprn_ctx* c; ~FtScope() { c->ft_s_now = nullptr; } } ft_scope{c};
        c->ft_s_now = (o.k0 == 0 && ft_fused) ? c->ft_s_phase : nullptr;
        if (o.k1 < c->T) side_stamp(o.k1, 6);          // (stamps of the NEXT panel's first step: behind the waits, behind "first")
        if (eager_J) { /* nothing left of "first" */ }
        else if (first_a_done == (int)J) {
            if ((rc = tiles(o.fb0, o.nfb, s1, shape_upd(o.nfb), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
        } else if ((rc = tiles(o.first0, o.nfirst, s1, shape_upd(o.nfirst), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
        if (o.k1 < c->T) side_stamp(o.k1, 7);
        // F_FIRST: by the first workgroup of the next launch on stream3 when that is a panel launch with the
        // synchronisation folded in (below), else a stream write
        if (eager_J) { /* no F_FIRST: nothing waits for it */ }
        else if (multi && folds_sync(o.k1)) pending_up = slot((int)J, F_FIRST) + 1;
        else HIP_TRY(c, raise(s1, (int)J, F_FIRST));
        if (o.nfirst > 0 && !eager_J) first_J = (int)J;
        // GPRN_FIRST_ALONE=n (experiments; default 0): with up to n matrices "next", "ahead" and "bulk" start behind "first"
        // instead of beside it.  stream3 cannot go on with the new panel before "first" is through -- the in-kernel stamps
        // (GPRN_STEP_STAMPS=2) show it there for 90-100 us with two matrices, the other launches taking the CUs at the same
        // moment -- and alone it takes 30-45 us; but the same launches then slow the panel products of the step behind it
        // (60 us instead of 30): 110.6 sweeps/s at config 3 either way
        static int first_alone = -1;
        if (first_alone < 0) { const char* e = getenv("GPRN_FIRST_ALONE"); first_alone = e ? atoi(e) : 0; }
        const int gate_kind = (use_flags && o.nfirst > 0 && nbatch <= first_alone && !eager_J) ? F_FIRST : F_PANEL;
        if (eager_J) { /* "next" is done */ }
        else {
        if (multi && sr && rest_J >= 0) {          // both waits of the "next" stream in one kernel
            FlagOps ops = {{nullptr, nullptr}, {slot((int)J, gate_kind) + 1, slot(rest_J, F_RESTA) + 1, nullptr, nullptr}};
            hipLaunchKernelGGL(k_flag_multi, dim3(1), dim3(64), 0, sn, ops, epoch, timed_out);
            HIP_TRY(c, hipGetLastError());
        } else {
        HIP_TRY(c, await(sn, (int)J, gate_kind));
        if (sr && rest_J >= 0) HIP_TRY(c, await(sn, rest_J, F_RESTA));
        }
        if (left) {
            for (int g = 1; g < o.k1 - o.k0 && o.k1 + g < c->T; ++g) {
                if ((rc = tiles(o.grp0[g], o.ngrp[g], sn, shape_upd(o.ngrp[g]), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
                HIP_TRY(c, raise(sn, o.k1 + g, F_NEXT));
                grp_pending[o.k1 + g] = 1;
                last_grp = o.k1 + g;
            }
        } else {
        if ((rc = tiles(o.next0, o.nnext, sn, shape_upd(o.nnext), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
        // (a stream write, not the launch's own end-of-kernel signal: with a fence and an atomic at the end of each of its
        // several hundred workgroups 116.6 vs 117.6 sweeps/s at config 3 with two matrices, 113.4 with six)
        HIP_TRY(c, raise(sn, (int)J, F_NEXT));
        if (o.nnext > 0) next_J = (int)J;
        }
        }   // !eager_J
        if (o.nrest) {
            // GPRN_BULK_SHAPE: workgroup shape of the bulk (TS_128x128 = 0: eight waves, two workgroups per
            // CU; TS_64x64 = 1: four per task, short-lived, leaves room on every CU for the chain's kernels)
            static int bulk_shape_small = -1, bulk_shape_big = -1, bulk_big_batch = -1;
            if (bulk_shape_small < 0) { const char* e = getenv("GPRN_BULK_SHAPE"); bulk_shape_small = e ? atoi(e) : TS_64x64; }
            // (experiments: another shape for phases of many matrices, which are bound by throughput, not by the chain)
            if (bulk_shape_big < 0) { const char* e = getenv("GPRN_BULK_SHAPE_BIG"); bulk_shape_big = e ? atoi(e) : bulk_shape_small; }
            if (bulk_big_batch < 0) { const char* e = getenv("GPRN_BULK_BIG_BATCH"); bulk_big_batch = e ? atoi(e) : 4; }
            const int bulk_shape = nbatch >= bulk_big_batch ? bulk_shape_big : bulk_shape_small;
            if (sr) {
                HIP_TRY(c, await(s2, (int)J, gate_kind));
                if ((rc = tiles(o.rest0, o.nrestA, s2, bulk_shape, GPRN_T_UPDATE_AHEAD, nosig, noaw, TG_AHEAD))) return rc;
                // F_RESTA: by the first workgroup of the "bulk" launch behind it (gprn_ctx::start_flag_now) when there is one
                const bool resta_by_bulk = multi && o.nrest > o.nrestA;
                if (resta_by_bulk) { c->start_flag_now = slot((int)J, F_RESTA) + 1; c->start_value_now = epoch; }
                else HIP_TRY(c, raise(s2, (int)J, F_RESTA));
                rc = tiles(o.rest0 + o.nrestA, o.nrest - o.nrestA, s2, bulk_shape, GPRN_T_UPDATE, nosig, noaw, TG_BULK);
                c->start_flag_now = nullptr;
                if (rc) return rc;
            } else if ((rc = tiles(o.rest0, o.nrest, s2, bulk_shape, GPRN_T_UPDATE, nosig, noaw, TG_BULK))) return rc;
            HIP_TRY(c, raise(s2, (int)J, F_REST));
            rest_J = (int)J;
        }
        // rows [k0, k1) of X are final once stream3 is through with the panel: their share of the phase's O(N^2)
        // reductions goes behind the panel's bulk update on the bulk stream (run_phase, api.hip)
        if (c->rows_final && !use_chain && !persist) {
            if (!(o.nrest && sr) && sn != s2) HIP_TRY(c, await(s2, (int)J, F_PANEL));
            if ((rc = c->rows_final(o.k0, o.k1, s2))) return rc;
            c->rows_done = o.k1;
            tail_on_s2 = true;
        }
        return GPRN_OK;
    };
    // behind the updates of the panel's last-but-one step: the early part of the panel's "first" outer update
    auto after_inner = [&](size_t J, int k) -> int {
        const gprn_ctx::OuterRange& o = c->outers[set][J];
        if (!split_first || use_chain || persist || k != o.k1 - 2 || o.nfa == 0 || o.nfirst == 0) return GPRN_OK;
        int r;
        if ((r = flush_inner())) return r;             // the chain's flag first: the launch below takes a while
        if (rest_J >= 0) HIP_TRY(c, await(s1, rest_J, sr_all ? F_RESTA : F_REST));   // same tiles as the previous panel's rest
        if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
        c->ft_s_now = (o.k0 == 0 && ft_fused) ? c->ft_s_phase : nullptr;
        r = tiles(o.fa0, o.nfa, s1, shape_upd(o.nfa), GPRN_T_PANEL, nosig, noaw, TG_NEXT);
        c->ft_s_now = nullptr;
        if (r) return r;
        first_a_done = (int)J;
        return GPRN_OK;
    };
    for (size_t J = 0; J < c->outers[set].size(); ++J) {
        const gprn_ctx::OuterRange& o = c->outers[set][J];
        for (int k = o.k0; k < o.k1; ++k) {
            const gprn_ctx::StepRange& s = c->steps[set][k];
            if (pending_outer >= 0 && (use_chain || s.npanel_l == 0) && (rc = do_outer(pending_outer))) return rc;
            if (use_chain && s.npanel_l > 0) {
                // stream3's half of the step; diag(k), L_{k+1,k} and B_{k+1,k+1} are k_chain's
                if (first_J >= 0) first_J = -1;        // (k_chain waits for F_FIRST itself)
                if ((rc = side_sync(k))) return rc;
                if ((rc = tiles(s.panel0 + 1, s.npanel_l - 1, s1, TS_64x128))) return rc;
                if (s.npanel > s.npanel_l) {
                    if ((rc = tiles(s.panel0 + s.npanel_l, s.npanel - s.npanel_l, s1, TS_128x64, GPRN_T_PANEL,
                                    x_part_then(k)))) return rc;
                } else HIP_TRY(c, await(s1, k, F_MINIL));
                if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
                if ((rc = tiles(s.upd0 + 1, s.nupd - 1, s1, shape_upd(s.nupd - 1)))) return rc;
                inner_k = k;
                continue;
            }
            if (use_chain) {                           // last tile step: diag(k) was k_chain's last act
                if ((rc = flush_inner())) return rc;
                if (k > 0) HIP_TRY(c, await(s0, k - 1, F_INNER));
                if ((rc = tiles(s.panel0, s.npanel, s0, TS_128x64))) return rc;
                continue;
            }
            // Every tile step looks the same to the chain, panel boundaries included (the two tiles the
            // next panel starts with are updated step by step, see ensure_tasks):
            //   chain  : diag(k)  ->  L_{k+1,k}  ->  B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T
            //   stream3: the other panel tiles, then the other in-panel updates of the step
            // two_streams: the diagonal blocks on the chain stream, the step's two tile launches on stream4, every
            // kernel waiting in-kernel for the flag of the one before it in the chain -- so each is dispatched
            // (arguments, task and pointer loads done, workgroups resident) while its predecessor still runs,
            // instead of after its completion has travelled through the stream
            const Await after_u = (two_streams && k > 0) ? in_kernel_wait(k - 1, F_U) : noaw;
            if (!persist && (rc = launch_diag(c, c->d_ptrs, nbatch, c->ld, k, c->d_info_cur, s0, in_kernel(k, F_DIAG), after_u))) return rc;
            if (!use_flags) HIP_TRY(c, raise(s0, k, F_DIAG));
            if (use_flags && k == 0 && c->chain_started) {
                // work handed over by the caller for the bulk stream (run_phase: the previous phase's X^T X
                // product, 528 long-running workgroups) goes behind the FIRST diagonal block: launched before
                // it, it holds every CU and the block waits for one to drain (211 us instead of 50 measured)
                HIP_TRY(c, await(s2, 0, F_DIAG));
                std::function<int()> f;
                f.swap(c->chain_started);
                if ((rc = f())) return rc;
            }
            if (s.npanel_l == 0) {
                // last tile step of the matrix: row k of the inverse is all that is left
                if ((rc = flush_inner())) return rc;
                // (its workgroups wait for stream3's last update themselves: a stream wait is a 5 us kernel of its own
                // on the chain stream, and the flag is up or about to be when they start; GPRN_LAST_WAIT=0: the stream wait)
                static int last_wait = -1;
                if (last_wait < 0) { const char* e = getenv("GPRN_LAST_WAIT"); last_wait = e ? atoi(e) : 1; }
                if (k > 0 && use_flags && last_wait && may_poll(2 * s.npanel * (size_t)nbatch)) {
                    if ((rc = tiles(s.panel0, s.npanel, s0, TS_128x64, GPRN_T_PANEL, nosig, in_kernel_wait(k - 1, F_INNER)))) return rc;
                    continue;
                }
                if (k > 0 && use_flags) {                  // (one wave waits, bounded by the budget like every in-kernel wait)
                    hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, s0, (unsigned*)nullptr, 0u,
                                       (const unsigned*)(slot(k - 1, F_INNER) + 1), epoch, timed_out);
                    HIP_TRY(c, hipGetLastError());
                } else if (k > 0) HIP_TRY(c, await(s0, k - 1, F_INNER));
                if ((rc = tiles(s.panel0, s.npanel, s0, TS_128x64))) return rc;
                continue;
            }
            // L_{k+1,k} reads what stream3's in-panel update of step k-1 wrote: in the flag schedule its
            // two workgroups per matrix poll that flag themselves (a stream wait is a 5 us kernel of its own)
            // ... when the chain is what bounds the phase (one or two matrices); with more, the phase is bound by the
            // tile kernels' throughput and 8 x batch resident 512-thread workgroups that only poll (80 us of every
            // loaded step, 139 VGPRs per lane) keep bulk workgroups off their CUs: a one-wave kernel waits instead.
            // GPRN_SPIN_MAX_BATCH overrides the limit (default 2).
            static int spin_max = -1;
            if (spin_max < 0) { const char* e = getenv("GPRN_SPIN_MAX_BATCH"); spin_max = e ? atoi(e) : 2; }
            const bool spin = use_flags && k > 0 && (nbatch <= spin_max || two_streams);
            if (use_flags && k > 0 && !spin) {
                hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, s0, (unsigned*)nullptr, 0u,
                                   (const unsigned*)(slot(k - 1, F_INNER) + 1), epoch, timed_out);
                HIP_TRY(c, hipGetLastError());
            }
            if (k > 0 && !use_flags) HIP_TRY(c, await(s0, k - 1, F_INNER));
            Await l_waits = spin ? in_kernel_wait(k - 1, F_INNER) : noaw;
            if (two_streams) {
                if (spin) { l_waits.flag2 = slot(k, F_DIAG) + 1; l_waits.value2 = epoch; }
                else l_waits = in_kernel_wait(k, F_DIAG);
            }
            hipStream_t sc = two_streams ? c->stream4 : s0;
            // (GPRN_CHAIN_ROWS=0: the throughput tile kernel for these two as well, as in round 1)
            static int chain_rows = -1;
            if (chain_rows < 0) { const char* e = getenv("GPRN_CHAIN_ROWS"); chain_rows = e ? atoi(e) : 1; }
            // GPRN_MINIL_BY_U=1 (default): L_{k+1,k}'s flag goes up at the START of the update launch behind it on the chain
            // stream instead of at the end of its own (launch_tile_rows) -- 1.7 us less between the two at every tile step;
            // stream3 sees the flag a launch gap later, at the end of a panel launch that runs ten times as long
            static int minil_by_u = -1;
            if (minil_by_u < 0) { const char* e = getenv("GPRN_MINIL_BY_U"); minil_by_u = e ? atoi(e) : 1; }
            const bool by_u = use_flags && chain_rows && !two_streams && minil_by_u;
            if (chain_rows) {
                if ((rc = launch_tile_rows(c, k, c->d_ptrs, nbatch, c->ld, 0, GPRN_T_PANEL, sc,
                                           by_u ? nosig : in_kernel(k, F_MINIL), l_waits))) return rc;
            } else if ((rc = tiles(s.panel0, 1, sc, TS_64x128, GPRN_T_PANEL, in_kernel(k, F_MINIL), l_waits))) return rc;
            if (!use_flags) HIP_TRY(c, raise(s0, k, F_MINIL));
            if (chain_rows) {
                if ((rc = launch_tile_rows(c, k, c->d_ptrs, nbatch, c->ld, 1, GPRN_T_PANEL, sc,
                                           two_streams ? in_kernel(k, F_U) : nosig, noaw,
                                           by_u ? slot(k, F_MINIL) + 1 : (unsigned*)nullptr, epoch))) return rc;
            } else if ((rc = tiles(s.upd0, 1, sc, TS_64x64, GPRN_T_PANEL, two_streams ? in_kernel(k, F_U) : nosig))) return rc;
            if (pending_outer >= 0 && (rc = do_outer(pending_outer))) return rc;    // the previous panel's trailing update
            // beside it: the rest of the panel, then the rest of the in-panel updates
            side_stamp(k, 0);
            const bool fold_sync = folds_sync(k);
            if (!fold_sync) {
                if (pending_up) { HIP_TRY(c, hipStreamWriteValue32(s1, pending_up, epoch, 0)); pending_up = nullptr; }
                if ((rc = side_sync(k))) return rc;
            }
            side_stamp(k, 1);
            if (merge_panel && tri && use_flags) {
                // (its last workgroup holds the launch open until L_{k+1,k} is there, see x_part_then)
                unsigned* up = nullptr;
                unsigned* up2 = nullptr;
                Await aw_d = noaw;
                if (fold_sync) {
                    if (inner_k >= 0 && !withheld(F_INNER)) up = slot(inner_k, F_INNER) + 1;
                    inner_k = -1;
                    up2 = pending_up;
                    pending_up = nullptr;
                    aw_d = in_kernel_wait(k, F_DIAG);
                }
                if ((rc = launch_panel(c, c->d_tasks + s.panel0 + 1, s.npanel_l - 1, s.npanel - s.npanel_l, c->d_ptrs,
                                       nbatch, c->ld, s1, x_part_then(k), aw_d, up, epoch, up2))) return rc;
                if (eager && o.k1 < c->T && c->esteps[k].ne > 0) {
                    // column k's share of the next panel's update, on the next-panel stream, as soon as column k of L and
                    // row k of X are there (the panel's first column also behind the previous panel's "ahead" launch,
                    // which wrote the same tiles)
                    const gprn_ctx::EStep& es = c->esteps[k];
                    FlagOps ops = {{nullptr, nullptr}, {slot(k, F_XW) + 1, slot(k, F_MINIL) + 1, nullptr, nullptr}};
                    if (k == o.k0 && rest_J >= 0) ops.wait[2] = slot(rest_J, F_RESTA) + 1;
                    hipLaunchKernelGGL(k_flag_multi, dim3(1), dim3(64), 0, c->stream4, ops, epoch, timed_out);
                    HIP_TRY(c, hipGetLastError());
                    c->ft_s_now = (k == 0 && ft_fused) ? c->ft_s_phase : nullptr;
                    rc = tiles(es.e0, es.ne, c->stream4, TS_64x64, GPRN_T_PANEL, nosig, noaw, TG_NEXT);
                    c->ft_s_now = nullptr;
                    if (rc) return rc;
                    if (k == o.k1 - 1) HIP_TRY(c, raise(c->stream4, (int)J, F_NEXT));
                }
            } else {
                if ((rc = tiles(s.panel0 + 1, s.npanel_l - 1, s1, TS_64x128))) return rc;
                if (s.npanel > s.npanel_l && use_flags) {
                    if ((rc = tiles(s.panel0 + s.npanel_l, s.npanel - s.npanel_l, s1, TS_128x64, GPRN_T_PANEL,
                                    x_part_then(k)))) return rc;
                } else {
                    if ((rc = tiles(s.panel0 + s.npanel_l, s.npanel - s.npanel_l, s1, TS_128x64))) return rc;
                    HIP_TRY(c, await(s1, k, F_MINIL));
                }
            }
            // (hundreds of workgroups: a fence + atomic in each would cost more than one stream write;
            // the flag goes up with stream3's next synchronisation kernel)
            // At the FIRST step of a panel the step's updates of the panel's other columns have to wait for the previous
            // panel's outer update ("next"), but the two tiles the chain's next step needs do not: they are the next
            // panel's eager tiles, kept up to date step by step and left out of the outer update (ensure_tasks).  Those two
            // go first, in a launch of their own that raises F_INNER itself, BEFORE the wait for "next": with one launch the
            // chain's L kernel sat 70-90 us at every panel boundary of a two-matrix phase waiting for an update it does not
            // read (profiles/r02_chain_timeline_cfg3.txt; the dataflow schedule of queue.hip made the false dependency
            // visible).  GPRN_SPLIT_INNER: 0 never, 1 (default) at panel boundaries, 2 at every step (one more launch per
            // step on stream3, itself a serial chain of launches: slower, 95.7 vs 103.2 sweeps/s in round 2).
            side_stamp(k, 2);                              // behind the panel
            static int split_inner = -1;
            if (split_inner < 0) { const char* e = getenv("GPRN_SPLIT_INNER"); split_inner = e ? atoi(e) : 1; }
            if (left) {
                // left-looking: column k+1 of the panel (and row k+1 of the right-hand side) with everything the panel has
                // produced so far; the previous panel's K = 512 update of THAT column / row is all that has to be there
                const gprn_ctx::LStep& ls = c->lsteps[k];
                const bool wait_grp = k + 1 < c->T && grp_pending[k + 1];
                static int split_inner_l = -1, split_max_l = -1;
                if (split_inner_l < 0) { const char* e = getenv("GPRN_SPLIT_INNER"); split_inner_l = e ? atoi(e) : 1; }
                if (split_max_l < 0) { const char* e = getenv("GPRN_SPLIT_INNER_MAX_BATCH"); split_max_l = e ? atoi(e) : 2; }
                if (use_flags && ls.ncrit > 0 && (split_inner_l >= 2 || (split_inner_l == 1 && wait_grp && k == o.k0 && nbatch <= split_max_l))) {
                    const bool skip = withheld(F_INNER);
                    if ((rc = tiles(ls.u0, ls.ncrit, s1, TS_64x64, GPRN_T_PANEL, skip ? nosig : in_kernel(k, F_INNER)))) return rc;
                    if (wait_grp) { HIP_TRY(c, await(s1, k + 1, F_NEXT)); grp_pending[k + 1] = 0; }
                    if ((rc = tiles(ls.u0 + ls.ncrit, ls.nu - ls.ncrit, s1, shape_upd(ls.nu - ls.ncrit)))) return rc;
                } else {
                    if (wait_grp) { HIP_TRY(c, await(s1, k + 1, F_NEXT)); grp_pending[k + 1] = 0; }
                    if ((rc = tiles(ls.u0, ls.nu, s1, shape_upd(ls.nu)))) return rc;
                    if (use_flags) inner_k = k;
                    else HIP_TRY(c, raise(s1, k, F_INNER));
                }
                if ((rc = after_inner(J, k))) return rc;
                continue;
            }
            const size_t ncrit = s.ncol1 > 0 ? s.ncol1 - 1 : 0;
            const bool boundary = next_J >= 0;             // this step's other updates wait for an outer update
            // (at panel boundaries only where the chain bounds the phase: one or two matrices -- config 2: +3 %; with six
            // matrices the phase is bound by throughput and the extra launch costs 1 %)
            static int split_max_batch = -1;
            if (split_max_batch < 0) { const char* e = getenv("GPRN_SPLIT_INNER_MAX_BATCH"); split_max_batch = e ? atoi(e) : 2; }
            if (use_flags && ncrit > 0 && (split_inner >= 2 || (split_inner == 1 && boundary && nbatch <= split_max_batch))) {
                const bool skip = withheld(F_INNER);
                if ((rc = tiles(s.upd0 + 1, ncrit, s1, TS_64x64, GPRN_T_PANEL, skip ? nosig : in_kernel(k, F_INNER)))) return rc;
                side_stamp(k, 3);
                if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
                side_stamp(k, 4);
                if ((rc = tiles(s.upd0 + 1 + ncrit, s.nupd - 1 - ncrit, s1, shape_upd(s.nupd - 1 - ncrit)))) return rc;
                side_stamp(k, 5);
            } else {
                if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
                side_stamp(k, 4);
                if ((rc = tiles(s.upd0 + 1, s.nupd - 1, s1, shape_upd(s.nupd - 1)))) return rc;
                side_stamp(k, 5);
                if (use_flags) inner_k = k;                // raised by stream3's next synchronisation kernel
                else HIP_TRY(c, raise(s1, k, F_INNER));    // an event wait sees only records made before it: the
                                                           // chain's wait for step k is enqueued at step k + 1
            }
            if ((rc = after_inner(J, k))) return rc;
        }
        if ((rc = flush_inner())) return rc;           // the chain's next step must not queue behind the outer update
        if (o.nfirst + o.nnext + o.nrest == 0) continue;
        // The outer update of this panel is ENQUEUED after the chain's three launches of the next panel's first step
        // (do_outer below): its dozen stream operations and launches take the host 60-100 us, during which the chain
        // stream ran dry at every panel boundary (profiles/r02_chain_timeline_cfg3.txt).
        pending_outer = (int)J;
    }
    if (pending_outer >= 0 && (rc = do_outer(pending_outer))) return rc;
    if (pending_up) { HIP_TRY(c, hipStreamWriteValue32(s1, pending_up, epoch, 0)); pending_up = nullptr; }
    if (tail_on_s2) HIP_TRY(c, raise(s2, 0, F_TAIL));
    static int multi_end = -1;
    if (multi_end < 0) { const char* e = getenv("GPRN_MULTI_FLAG"); multi_end = e ? atoi(e) : 1; }
    if (use_flags && multi_end && !(left && last_grp >= 0)) {
        // the chain stream joins the others in ONE kernel (up to four stream waits before)
        FlagOps ops = {{nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
        int nw = 0;
        if (first_J >= 0) ops.wait[nw++] = slot(first_J, F_FIRST) + 1;
        if (next_J >= 0) ops.wait[nw++] = slot(next_J, F_NEXT) + 1;
        if (rest_J >= 0) ops.wait[nw++] = slot(rest_J, F_REST) + 1;
        if (tail_on_s2) ops.wait[nw++] = slot(0, F_TAIL) + 1;
        if (nw) {
            hipLaunchKernelGGL(k_flag_multi, dim3(1), dim3(64), 0, s0, ops, epoch, timed_out);
            HIP_TRY(c, hipGetLastError());
        }
        return GPRN_OK;
    }
    if (first_J >= 0) HIP_TRY(c, await(s0, first_J, F_FIRST));
    if (next_J >= 0) HIP_TRY(c, await(s0, next_J, F_NEXT));
    if (left && last_grp >= 0) HIP_TRY(c, await(s0, last_grp, F_NEXT));
    if (rest_J >= 0) HIP_TRY(c, await(s0, rest_J, F_REST));
    if (tail_on_s2) HIP_TRY(c, await(s0, 0, F_TAIL));
    return GPRN_OK;
}

// Block schedule.  The latency chain factors and inverts the 512 x 512 diagonal block D of an outer panel
// entirely on its own stream -- per tile step the diagonal tile, the in-block panel and the in-block update, nothing of
// which waits for another stream -- and the rest of the panel is not done step by step at all:
//     L[i, panel] = B[i, panel] X_D^T  (i >= k1)       X[panel, c] = X_D R[panel, c]  (c < k0)
// are ONE launch each per panel (K <= 512, into transposed mirrors, see ensure_tasks), followed by the K = 512 trailing
// update in four classes: the next panel's diagonal block (the chain's own launch: its next input), the rest of the next
// panel's columns / rows, the panel after next's, everything beyond.  Against factor_invert_split: no step-synchronous
// side stream (its three dependent launches per tile step were what the chain waited for when one or two matrices are
// factored), a tenth of the launches and stream operations, and the in-panel tiles are read and written once per panel
// instead of up to three times with K = 128.
//   chain   : per step  diag(k) -> in-block panel -> in-block update;  per panel  [X_D complete: F_PANEL], wait F_NEXT(P-1),
//             the L mirrors of the next block's rows [F_FIRST] -> update of the next diagonal block
//   stream3 : wait F_PANEL(P), F_NEXT(P-1);  L mirrors of the other rows [F_MINIL]
//   stream4 : wait F_PANEL(P), F_NEXT(P-1);  X mirrors + copy back [F_XW];  wait F_MINIL(P), F_FIRST(P), F_RESTA(P-1);  "next" [F_NEXT]
//   bulk    : wait F_MINIL(P), F_XW(P);  "ahead" [F_RESTA], "bulk" [F_REST], the phase's row reductions (rows_final)
static int factor_invert_blocks(gprn_ctx* c, int nbatch)
{
    int rc;
    hipStream_t s0 = c->stream, s1 = c->stream3, s2 = c->stream2, s4 = c->stream4;
    const int use_flags = factor_use_flags(c);
    enum { F_DIAG = 0, F_MINIL, F_INNER, F_PANEL, F_NEXT, F_REST, F_FIRST, F_XW, F_U, F_RESTA, F_TAIL, F_KINDS };
    static_assert(F_KINDS == GPRN_FLAG_KINDS, "flag kinds");
    if (use_flags && c->sig_T < c->T) {
        if (c->d_sig) hipFree(c->d_sig);
        c->d_sig = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_sig, ((size_t)c->T * F_KINDS * 2 + 4) * sizeof(unsigned)));
        HIP_TRY(c, hipMemset(c->d_sig, 0, ((size_t)c->T * F_KINDS * 2 + 4) * sizeof(unsigned)));
        c->sig_T = c->T;
        c->epoch = 0;
        c->sig_budget_ms = -1;
    }
    if (use_flags && c->sig_budget_ms != c->wait_budget_ms) {
        const unsigned ticks = (unsigned)std::min<long long>(0xffffffffll, (long long)c->wait_budget_ms * 100000ll);
        HIP_TRY(c, hipMemcpy(c->d_sig + (size_t)c->sig_T * F_KINDS * 2 + 1, &ticks, sizeof(unsigned), hipMemcpyHostToDevice));
        c->sig_budget_ms = c->wait_budget_ms;
    }
    const unsigned epoch = ++c->epoch;
    hipEvent_t events[F_KINDS] = {c->ev_diag, c->ev_minil, c->ev_inner, c->ev_panel, c->ev_next, c->ev_rest, c->ev_first, c->ev_xw, nullptr, c->ev_resta, c->ev_tail};
    auto slot = [&](int idx, int kind) { return c->d_sig + ((size_t)idx * F_KINDS + kind) * 2; };
    unsigned* timed_out = c->d_sig ? c->d_sig + (size_t)c->sig_T * F_KINDS * 2 : nullptr;
    auto raise = [&](hipStream_t st, int idx, int kind) {
        return use_flags ? hipStreamWriteValue32(st, slot(idx, kind) + 1, epoch, 0) : hipEventRecord(events[kind], st);
    };
    auto await = [&](hipStream_t st, int idx, int kind) {
        return use_flags ? hipStreamWaitValue32(st, slot(idx, kind) + 1, epoch, hipStreamWaitValueGte, 0xffffffffu)
                         : hipStreamWaitEvent(st, events[kind], 0);
    };
    auto in_kernel = [&](int idx, int kind) {      // the launch raises the flag itself (flag schedule)
        return use_flags ? Signal{slot(idx, kind), epoch, nullptr, 0, timed_out} : Signal{nullptr, 0, nullptr, 0, nullptr};
    };
    // the chain waits through a one-thread kernel (bounded by the wait budget) instead of a stream wait
    auto chain_wait = [&](int idx, int kind) -> int {
        if (!use_flags) { HIP_TRY(c, hipStreamWaitEvent(s0, events[kind], 0)); return GPRN_OK; }
        hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, s0, (unsigned*)nullptr, 0u,
                           (const unsigned*)(slot(idx, kind) + 1), epoch, timed_out);
        HIP_TRY(c, hipGetLastError());
        return GPRN_OK;
    };
    const Await noaw{nullptr, 0, nullptr, nullptr, 0};
    const Signal nosig{nullptr, 0, nullptr, 0, nullptr};
    static size_t big = 0;                         // tasks x batch above which 128x128 workgroups pay
    if (!big) { const char* e = getenv("GPRN_FEW_TASKS"); big = e && atoi(e) > 0 ? (size_t)atoi(e) : 4000; }
    auto shape_for = [&](size_t n) { return n * (size_t)nbatch > big ? TS_128x128 : TS_64x64; };
    auto tiles = [&](size_t first, size_t n, hipStream_t st, int shape, int fam, int tag, Signal sig = Signal{nullptr, 0, nullptr, 0, nullptr}) {
        return launch_tiles(c, c->d_tasks + first, n, c->d_ptrs, nbatch, c->ld, fam, st, shape, sig, Await{nullptr, 0, nullptr, nullptr, 0}, tag);
    };
    static int bulk_shape = -1;
    if (bulk_shape < 0) { const char* e = getenv("GPRN_BULK_SHAPE"); bulk_shape = e ? atoi(e) : TS_64x64; }
    int first_raises = 0;
    auto withheld = [&]() {                        // test hook (gprn_set_option "withhold_inner"): the n-th F_FIRST signal
        return use_flags && c->withhold_inner > 0 && ++first_raises == c->withhold_inner;
    };
    if (!use_flags && c->chain_started) {          // event schedule: nothing to gate it on
        std::function<int()> f;
        f.swap(c->chain_started);
        if ((rc = f())) return rc;
    }
    const int NP = (int)c->bpanels.size();
    bool tail_on_s2 = false;
    int last_next = -1, last_rest = -1, last_x = -1;
    for (int P = 0; P < NP; ++P) {
        const gprn_ctx::BlkPanel& bp = c->bpanels[P];
        // ---- the chain: the diagonal block of the panel
        for (int k = bp.k0; k < bp.k1; ++k) {
            const gprn_ctx::BlkStep& b = c->bsteps[k];
            const bool hook = use_flags && k == 0 && (bool)c->chain_started;
            if ((rc = launch_diag(c, c->d_ptrs, nbatch, c->ld, k, c->d_info_cur, s0, hook ? in_kernel(0, F_DIAG) : nosig, noaw))) return rc;
            if (hook) {
                // work handed over by the caller for the bulk stream goes behind the FIRST diagonal block (see factor_invert_split)
                HIP_TRY(c, await(s2, 0, F_DIAG));
                std::function<int()> f;
                f.swap(c->chain_started);
                if ((rc = f())) return rc;
            }
            // in-block panel: L_ik (k < i < k1), X_kc (k0 <= c < k); the last step of the panel completes X_D
            if (b.nl && (rc = launch_panel_rows(c, c->d_tasks + b.l0, b.nl_l, b.nl - b.nl_l, c->d_ptrs, nbatch, c->ld, s0, nosig, noaw))) return rc;
            // (GPRN_BLK_U=0: the throughput kernel for the in-block updates, experiments)
            static int blk_u = -1;
            if (blk_u < 0) { const char* e = getenv("GPRN_BLK_U"); blk_u = e ? atoi(e) : 1; }
            if (b.nu && blk_u && (rc = launch_blk_update(c, c->d_tasks + b.u0, b.nu, c->d_ptrs, nbatch, c->ld, s0, nosig, noaw))) return rc;
            if (b.nu && !blk_u && (rc = tiles(b.u0, b.nu, s0, TS_64x64, GPRN_T_PANEL, TG_INNER))) return rc;
        }
        HIP_TRY(c, raise(s0, P, F_PANEL));                           // L_D, X_D are complete
        if (bp.ntl) {
            // ---- the chain goes on with the L mirrors of the next block's rows and the update of the next diagonal
            // block.  B[i, panel] and that block were last written by the previous panel's "next" update.
            if (last_next >= 0 && (rc = chain_wait(last_next, F_NEXT))) return rc;
            // test hook ("withhold_inner" = n): the chain's n-th boundary waits for a flag nobody raises -- the in-kernel
            // wait gives up at its budget and the call is re-run on events (F_U is not used by this schedule)
            if (withheld() && (rc = chain_wait(0, F_U))) return rc;
            if (use_flags) {
                if ((rc = tiles(bp.tl0, bp.ntl_early, s0, TS_64x64, GPRN_T_PANEL, TG_TRMM, in_kernel(P, F_FIRST)))) return rc;
            } else {
                if ((rc = tiles(bp.tl0, bp.ntl_early, s0, TS_64x64, GPRN_T_PANEL, TG_TRMM))) return rc;
                HIP_TRY(c, raise(s0, P, F_FIRST));
            }
            if ((rc = tiles(bp.dn0, bp.ndn, s0, TS_64x64, GPRN_T_PANEL, TG_NEXT))) return rc;
            // ---- stream3: the L mirrors of the other rows
            HIP_TRY(c, await(s1, P, F_PANEL));
            if (last_next >= 0) HIP_TRY(c, await(s1, last_next, F_NEXT));
            if ((rc = tiles(bp.tl0 + bp.ntl_early, bp.ntl - bp.ntl_early, s1, shape_for(bp.ntl - bp.ntl_early), GPRN_T_PANEL, TG_TRMM))) return rc;
            HIP_TRY(c, raise(s1, P, F_MINIL));
        }
        // ---- stream4: the X mirrors and their copy back; then the "next" part of the update
        HIP_TRY(c, await(s4, P, F_PANEL));
        if (last_next >= 0 && last_next != P - 1) HIP_TRY(c, await(s4, last_next, F_NEXT));   // (its own "next" is in order)
        if (bp.ntx) {
            if ((rc = tiles(bp.tx0, bp.ntx, s4, shape_for(bp.ntx), GPRN_T_PANEL, TG_TRMM))) return rc;
            if ((rc = launch_tcopy(c, c->d_tasks + bp.cbx0, bp.ntx, c->d_ptrs, nbatch, c->ld, s4))) return rc;
        }
        HIP_TRY(c, raise(s4, P, F_XW));
        last_x = P;
        if (bp.ntl) {
            // ---- "next" on stream4, "ahead" / "bulk" on the bulk stream
            HIP_TRY(c, await(s4, P, F_MINIL));
            HIP_TRY(c, await(s4, P, F_FIRST));
            if (last_rest >= 0) HIP_TRY(c, await(s4, last_rest, F_RESTA));
            if ((rc = tiles(bp.next0, bp.nnext, s4, shape_for(bp.nnext), GPRN_T_PANEL, TG_NEXT))) return rc;
            HIP_TRY(c, raise(s4, P, F_NEXT));
            last_next = P;
            HIP_TRY(c, await(s2, P, F_MINIL));
            HIP_TRY(c, await(s2, P, F_XW));
            if ((rc = tiles(bp.rest0, bp.nrestA, s2, bulk_shape, GPRN_T_UPDATE_AHEAD, TG_AHEAD))) return rc;
            HIP_TRY(c, raise(s2, P, F_RESTA));
            if ((rc = tiles(bp.rest0 + bp.nrestA, bp.nrest - bp.nrestA, s2, bulk_shape, GPRN_T_UPDATE, TG_BULK))) return rc;
            HIP_TRY(c, raise(s2, P, F_REST));
            last_rest = P;
            // rows [k0, k1) of X are final (in-block tiles by the chain, the others copied back): their share of the
            // phase's O(N^2) reductions, behind the panel's bulk update
            if (c->rows_final) {
                if ((rc = c->rows_final(bp.k0, bp.k1, s2))) return rc;
                c->rows_done = bp.k1;
                tail_on_s2 = true;
            }
        }
    }
    // ---- joins; callers that want L itself (and clean upper triangles) get the mirrors copied into place
    if (last_x >= 0) HIP_TRY(c, await(s0, last_x, F_XW));
    if (last_next >= 0) HIP_TRY(c, await(s0, last_next, F_NEXT));
    if (last_rest >= 0) HIP_TRY(c, await(s0, last_rest, F_REST));
    if (tail_on_s2) {
        HIP_TRY(c, raise(s2, 0, F_TAIL));
        HIP_TRY(c, await(s0, 0, F_TAIL));
    }
    if (!c->fast_factor)
        for (int P = 0; P < NP; ++P) {
            const gprn_ctx::BlkPanel& bp = c->bpanels[P];
            if ((rc = launch_tcopy(c, c->d_tasks + bp.cbl0, bp.ntl, c->d_ptrs, nbatch, c->ld, s0))) return rc;
        }
    return GPRN_OK;
}

// a dependency wait inside a chain kernel gave up (see Await): the results of that call are void
int factor_check_waits(gprn_ctx* c)
{
    if (c->d_stamps && c->stamps_T >= c->T && c->T > 1) {     // development aid: print the last chain's timeline
        std::vector<unsigned long long> h((size_t)c->T * 8);
        if (hipMemcpy(h.data(), c->d_stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost) == hipSuccess) {
            double sum[8] = {0};
            int n = 0;
            for (int k = 1; k + 2 < c->T; ++k, ++n)
                for (int i = 0; i < 8; ++i) {
                    const unsigned long long a = h[(size_t)k * 8 + i], b = i < 7 ? h[(size_t)k * 8 + i + 1] : h[(size_t)(k + 1) * 8];
                    sum[i] += (double)(b - a) * 0.01;        // 100 MHz ticks -> us
                }
            if (n > 0)
                fprintf(stderr, "[gprn] chain us/step over %d steps: diag %.1f publish %.1f -> %.1f wait %.1f L %.1f publish %.1f "
                                "U %.1f drain %.1f\n", n, sum[0] / n, sum[1] / n, sum[2] / n, 0.0, sum[3] / n, sum[4] / n,
                        sum[5] / n, sum[6] / n + sum[7] / n);
        }
    }
    if (c->d_step_stamps && c->step_stamps_n > 0) {            // development aid: the last factorisations' chains as they ran
        const int T = c->step_stamps_T, nph = std::min(c->step_stamps_n, 8);
        std::vector<unsigned long long> h((size_t)8 * T * 9), side;
        if (c->d_side_stamps) {
            side.resize((size_t)T * 8);
            if (hipMemcpy(side.data(), c->d_side_stamps, side.size() * sizeof(side[0]), hipMemcpyDeviceToHost) != hipSuccess) side.clear();
        }
        if (hipMemcpy(h.data(), c->d_step_stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int back = std::min(nph, 2); back >= 1; --back) {
                const int ph = (c->step_stamps_n - back) & 7;
                const unsigned long long* p = h.data() + (size_t)ph * T * 9;
                if (!p[0]) continue;
                fprintf(stderr, "[gprn] chain of factorisation %d (batch %d), us: step | diag: ->start run | L: ->launch wait run | U: ->launch run | step total\n",
                        c->step_stamps_n - back, c->step_stamps_batch[ph]);
                unsigned long long prev_end = p[0];
                for (int k = 0; k < c->T; ++k) {
                    const unsigned long long* d = p + (size_t)k * 9;
                    const unsigned long long* l = d + 3;
                    const unsigned long long* u = d + 6;
                    if (!d[0]) break;
                    auto us = [](unsigned long long a, unsigned long long b) { return b >= a ? (double)(b - a) * 0.01 : -1.0; };
                    if (l[0] && u[0]) {
                        fprintf(stderr, "  %3d | %6.1f %6.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f | %7.1f", k, us(prev_end, d[0]), us(d[0], d[2]),
                                us(d[2], l[0]), us(l[0], l[1]), us(l[1], l[2]), us(l[2], u[0]), us(u[0], u[2]), us(prev_end, u[2]));
                        if (c->d_side_stamps && ph == c->side_stamps_ph && side.size()) {
                            // stream3, relative to the END of this step's diagonal block: before its synchronisation kernel,
                            // behind it, behind the panel, behind the chain's two tiles, behind the other updates
                            const unsigned long long* sd = side.data() + (size_t)k * 8;
                            fprintf(stderr, "   s3:");
                            for (int i = 0; i < 8; ++i)
                                if (sd[i]) fprintf(stderr, " %7.1f", sd[i] >= d[2] ? (double)(sd[i] - d[2]) * 0.01 : -(double)(d[2] - sd[i]) * 0.01);
                                else fprintf(stderr, "       -");
                        }
                        fprintf(stderr, "\n");
                    }
                    else
                        fprintf(stderr, "  %3d | %6.1f %6.1f |\n", k, us(prev_end, d[0]), us(d[0], d[2]));
                    prev_end = u[0] ? u[2] : d[2];
                }
            }
        }
        c->step_stamps_n = 0;
    }
    {
        const int rq = queue_check_waits(c);
        if (rq) return rq;
    }
    if (!c->d_sig) return GPRN_OK;
    unsigned word[3] = {0, 0, 0};                      // sticky word, budget, which flag (spin_until)
    HIP_TRY(c, hipMemcpy(word, c->d_sig + (size_t)c->sig_T * GPRN_FLAG_KINDS * 2, sizeof(word), hipMemcpyDeviceToHost));
    if (word[0]) {
        hipMemset(c->d_sig + (size_t)c->sig_T * GPRN_FLAG_KINDS * 2, 0, sizeof(unsigned));
        static const char* const kind_name[GPRN_FLAG_KINDS] = {"DIAG", "MINIL", "INNER", "PANEL", "NEXT", "REST", "FIRST", "XW", "U", "RESTA", "TAIL"};
        const long long at = (long long)c->sig_T * GPRN_FLAG_KINDS * 2 + (long long)(int)word[2];
        char what[96];
        if (at >= 0 && at < (long long)c->sig_T * GPRN_FLAG_KINDS * 2)
            snprintf(what, sizeof(what), " (flag %s of tile step / panel %lld, T = %d)", kind_name[(at / 2) % GPRN_FLAG_KINDS],
                     at / 2 / GPRN_FLAG_KINDS, c->T);
        else snprintf(what, sizeof(what), " (a flag outside the factorisation's table)");
        c->err = std::string("factorisation: a device-side dependency wait timed out") + what;
        c->last_timeout = what;
        return GPRN_E_WAIT_TIMEOUT;
    }
    return GPRN_OK;
}

static int factor_invert_impl(gprn_ctx* c, int nbatch);

// GPRN_TIME_ENQUEUE=1 (probes): host time spent enqueueing factorisations, printed every 64 calls
int factor_invert(gprn_ctx* c, int nbatch)
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("GPRN_TIME_ENQUEUE"); on = e ? atoi(e) : 0; }
    if (!on) return factor_invert_impl(c, nbatch);
    static double total = 0.0;
    static int calls = 0;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = factor_invert_impl(c, nbatch);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    total += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    if (++calls % 64 == 0) fprintf(stderr, "[gprn] host enqueue of a factorisation: %.3f ms on average over %d calls (T = %d, batch %d)\n", total / calls, calls, c->T, nbatch);
    return rc;
}

static int factor_invert_impl(gprn_ctx* c, int nbatch)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    static int lat_max = 0;                        // GPRN_LAT_MAX overrides (experiments)
    if (!lat_max) { const char* e = getenv("GPRN_LAT_MAX"); lat_max = e && atoi(e) > 0 ? atoi(e) : 32; }
    static int step_stamps_env = -1;
    if (step_stamps_env < 0) { const char* e = getenv("GPRN_STEP_STAMPS"); step_stamps_env = e ? atoi(e) : 0; }
    c->side_stamps = nullptr;
    if (step_stamps_env) {
        if (c->step_stamps_T < c->T) {
            if (c->d_step_stamps) hipFree(c->d_step_stamps);
            if (c->d_side_stamps) hipFree(c->d_side_stamps);
            c->d_side_stamps = nullptr;
            HIP_TRY(c, hipMalloc(&c->d_side_stamps, (size_t)c->T * 8 * sizeof(unsigned long long)));
            HIP_TRY(c, hipMalloc(&c->d_step_stamps, (size_t)8 * c->T * 9 * sizeof(unsigned long long)));
            c->step_stamps_T = c->T;
            c->step_stamps_n = 0;
        }
        c->step_stamps_n += 1;
        const int ph = (c->step_stamps_n - 1) & 7;
        c->step_stamps_batch[ph] = nbatch;
        HIP_TRY(c, hipMemsetAsync(c->d_step_stamps + (size_t)ph * c->step_stamps_T * 9, 0, (size_t)c->step_stamps_T * 9 * sizeof(unsigned long long), c->stream));
        if (step_stamps_env >= 2 && nbatch <= 2) {       // (the node phase: the one the chain bounds)
            HIP_TRY(c, hipMemsetAsync(c->d_side_stamps, 0, (size_t)c->step_stamps_T * 8 * sizeof(unsigned long long), c->stream));
            c->side_stamps = c->d_side_stamps;
            c->side_stamps_ph = ph;
        }
    }
    if (!(split_sched() && !(queue_enabled(c) && c->T > 1))) c->rows_final = nullptr;   // only the launch schedule calls it
    // B still to be built (run_phase): the launch schedule builds what its first panel touches and forms the rest inside
    // that panel's update (factor_invert_split); every other schedule gets all of it now
    auto build_all_now = [&]() -> int {
        const int pend = c->build_pending;
        c->build_pending = 0;
        return pend ? vec_build_B(c, pend) : GPRN_OK;
    };
    if (split_sched() && queue_enabled(c) && c->T > 1) {
        if ((rc = build_all_now())) return rc;
        rc = factor_invert_queue(c, nbatch, nbatch * c->T <= lat_max ? 1 : 0);
        if (rc) {
            // an enqueue that broke off half-way leaves kernels polling for nodes nobody will finish: end their waits
            // (the results are void, the caller gets the error), then clear the verdict that belongs to this call
            if (c->d_qctr) {
                const unsigned one = 1u;
                unsigned* const tmo = c->d_qctr + QC_TIMEOUT * GPRN_QCTR_STRIDE;
                (void)hipMemcpy(tmo, &one, sizeof(unsigned), hipMemcpyHostToDevice);
                (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream2);
                (void)hipStreamSynchronize(c->stream3);
                (void)hipMemset(tmo, 0, sizeof(unsigned));
            }
            c->q_lauum.n = 0;
        }
        return rc;
    }
    // GPRN_BLOCK_SCHED / option "block_sched": 1 = the block schedule where it applies (throughput set, at least three
    // outer panels), 0 (default) = the step-synchronous launch schedule everywhere.  Measured at config 3: 103.7-104.5
    // sweeps/s with it in both phases, 107-108.5 in the weight phase only, against 110 without
    static int blk_env = -1;
    if (blk_env < 0) { const char* e = getenv("GPRN_BLOCK_SCHED"); blk_env = e ? atoi(e) : 0; }
    const int blk = c->block_sched >= 0 ? c->block_sched : blk_env;
    static int blk_min = -1, blk_max = -1;         // GPRN_BLOCK_MIN_BATCH / GPRN_BLOCK_MAX_BATCH (experiments): batches it applies to
    if (blk_min < 0) { const char* e = getenv("GPRN_BLOCK_MIN_BATCH"); blk_min = e ? atoi(e) : 0; }
    if (blk_max < 0) { const char* e = getenv("GPRN_BLOCK_MAX_BATCH"); blk_max = e ? atoi(e) : 1 << 30; }
    const bool blocks = split_sched() && blk && !(nbatch * c->T <= lat_max) && c->bpanels.size() >= 3 && c->stream4 &&
                        nbatch >= blk_min && nbatch <= blk_max;
    static int chain_env_b = -1;                   // (the persistent chain kernels start before anything else is enqueued)
    if (chain_env_b < 0) { const char* e = getenv("GPRN_CHAIN"); chain_env_b = e ? atoi(e) : 0; }
    if ((blocks || !split_sched() || chain_env_b != 0) && (rc = build_all_now())) return rc;
    if (blocks) {
        static int side_pad = -1;                  // GPRN_BLK_SIDE_PAD=0: no LDS pad on the side streams' launches
        if (side_pad < 0) { const char* e = getenv("GPRN_BLK_SIDE_PAD"); side_pad = e ? atoi(e) : 1; }
        c->pad_side_now = side_pad != 0;
        rc = factor_invert_blocks(c, nbatch);
        c->pad_side_now = false;
    } else if (split_sched()) {
        // GPRN_SIDE_PAD_MAX_BATCH (experiments): up to that many matrices every launch off the chain stream carries the
        // small-batch LDS pad (one tile workgroup per CU), not only the bulk
        static int side_pad_max = -1;
        if (side_pad_max < 0) { const char* e = getenv("GPRN_SIDE_PAD_MAX_BATCH"); side_pad_max = e ? atoi(e) : 0; }
        c->pad_side_now = nbatch <= side_pad_max;
        rc = factor_invert_split(c, nbatch, nbatch * c->T <= lat_max ? 1 : 0);
        c->pad_side_now = false;
    }
    if (split_sched()) {
        if (rc && c->d_sig && c->use_flags == 1) {
            // The enqueue broke off half-way: stream waits already queued on the device's shared streams would
            // wait for flags nobody will raise (they have no time-out).  Put every flag of this call up so that
            // the streams drain; the results are void, the caller gets the error.
            std::vector<unsigned> h((size_t)c->sig_T * GPRN_FLAG_KINDS * 2, 0u);
            for (size_t i = 1; i < h.size(); i += 2) h[i] = c->epoch;
            (void)hipMemcpy(c->d_sig, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream2);
            (void)hipStreamSynchronize(c->stream3);
            if (c->stream4) (void)hipStreamSynchronize(c->stream4);
            // (a wait may have given up before the flags went up: that verdict belongs to this failed call)
            (void)hipMemset(c->d_sig + (size_t)c->sig_T * GPRN_FLAG_KINDS * 2, 0, sizeof(unsigned));
        }
        return rc;
    }
    bool rest_pending = false, next_pending = false;
    const Signal nosig{nullptr, 0, nullptr, 0, nullptr};
    const Await noaw{nullptr, 0, nullptr};
    // Launches with few tasks are latency-bound (one workgroup per 128x128 task, K = 128 or 512 of
    // serial MFMA work each): cut their tasks into 64-row / 64-column pieces to use the idle CUs.
    // In-place panel tasks may only be cut along the dimension they do not read across.
    static size_t few_max = 0;                     // GPRN_FEW_TASKS overrides (experiments)
    if (!few_max) { const char* e = getenv("GPRN_FEW_TASKS"); few_max = e && atoi(e) > 0 ? (size_t)atoi(e) : 4000; }
    auto few = [&](size_t ntasks) { return ntasks * (size_t)nbatch <= few_max; };
    const int set = nbatch * c->T <= 32 ? 1 : 0;   // latency schedule: little work in total (measured:
                                                   // +11 % at N=2048 x 1 matrix, -3 % at N=4096 x 2)
    if (c->chain_started) {
        std::function<int()> f;
        f.swap(c->chain_started);
        if ((rc = f())) return rc;
    }
    for (size_t J = 0; J < c->outers[set].size(); ++J) {
        const gprn_ctx::OuterRange& o = c->outers[set][J];
        for (int k = o.k0; k < o.k1; ++k) {            // the latency chain of this panel
            const gprn_ctx::StepRange& s = c->steps[set][k];
            if ((rc = launch_diag(c, c->d_ptrs, nbatch, c->ld, k, c->d_info_cur))) return rc;
            if (few(s.npanel)) {
                if ((rc = launch_tiles(c, c->d_tasks + s.panel0, s.npanel_l, c->d_ptrs, nbatch, c->ld,
                                       GPRN_T_PANEL, nullptr, TS_64x128, nosig, noaw, TG_PANEL))) return rc;
                if ((rc = launch_tiles(c, c->d_tasks + s.panel0 + s.npanel_l, s.npanel - s.npanel_l,
                                       c->d_ptrs, nbatch, c->ld, GPRN_T_PANEL, nullptr, TS_128x64, nosig, noaw, TG_PANEL))) return rc;
            } else if ((rc = launch_tiles(c, c->d_tasks + s.panel0, s.npanel, c->d_ptrs, nbatch, c->ld,
                                          GPRN_T_PANEL, nullptr, TS_128x128, nosig, noaw, TG_PANEL))) return rc;
            if (next_pending) {                        // the other columns / rows of this panel
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_next, 0));
                next_pending = false;
            }
            if ((rc = launch_tiles(c, c->d_tasks + s.upd0, s.nupd, c->d_ptrs, nbatch, c->ld,
                                   GPRN_T_PANEL, nullptr, few(s.nupd) ? TS_64x64 : TS_128x128, nosig, noaw, TG_INNER))) return rc;
        }
        if (o.nfirst + o.nnext + o.nrest == 0) continue;
        // Outer update of panel J.  On the chain stream only what the next panel's first tile step
        // needs (its first column of B, first row of R); the rest of the next panel and everything
        // beyond go to the second stream, which the chain joins before its first in-panel update
        // (the first kernel that touches those tiles again).
        HIP_TRY(c, hipEventRecord(c->ev_panel, c->stream));
        if (rest_pending)                              // same tiles as the previous panel's rest
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_rest, 0));
        if ((rc = launch_tiles(c, c->d_tasks + o.first0, o.nfirst, c->d_ptrs, nbatch, c->ld,
                               GPRN_T_PANEL, nullptr, few(o.nfirst) ? TS_64x64 : TS_128x128, nosig, noaw, TG_NEXT))) return rc;
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_panel, 0));
        if ((rc = launch_tiles(c, c->d_tasks + o.next0, o.nnext, c->d_ptrs, nbatch, c->ld,
                               GPRN_T_PANEL, c->stream2, few(o.nnext) ? TS_64x64 : TS_128x128, nosig, noaw, TG_NEXT))) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_next, c->stream2));
        next_pending = o.nnext > 0;
        if (o.nrest) {
            if ((rc = launch_tiles(c, c->d_tasks + o.rest0, o.nrest, c->d_ptrs, nbatch, c->ld,
                                   GPRN_T_UPDATE, c->stream2, TS_128x128, nosig, noaw, TG_BULK))) return rc;
            HIP_TRY(c, hipEventRecord(c->ev_rest, c->stream2));
            rest_pending = true;
        }
    }
    if (next_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_next, 0));
    if (rest_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_rest, 0));
    return GPRN_OK;
}

// BUF_B of every slot = lower(X^T X), X in BUF_X (L in BUF_B is overwritten)
int lauum_lower(gprn_ctx* c, int nbatch, hipStream_t stream)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    return launch_tiles(c, c->d_tasks + c->lauum0, c->nlauum, c->d_ptrs, nbatch, c->ld,
                        GPRN_T_LAUUM, stream);
}
