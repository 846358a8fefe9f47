// The small-N path: problems of one tile (N <= 128; two tiles on request), the regime the reference's own examples and its
// optimiser / sampler loops live in (docs/examples/one_dataset.ipynb: N = 45, "took 2.79 ms" per nELBO; meanfield.py:1095-1152,
// 1222-1260).  There a sweep of the launch schedule is ~25 launches of one workgroup each and the time is all dispatch.
// Here a half-sweep is ONE launch, one workgroup per latent GP:
//
//   k_small_phase   d, s, pred, z  ->  B = I + D^1/2 K D^1/2  ->  B = L L^T, X = L^-1 (diag_tile; for two tiles the 2 x 2
//                   blocked form with the tile contraction on the workgroup's four waves)  ->  u = X z, column sums of X
//                   ->  new mu, var, tr B^-1, log det B                         meanfield.py:759-792 / 838-865, 1087-1091
//   k_small_tail    per latent GP  mu^T K^-1 mu = |L_K^-1 m|^2 (:1032,1050);  per node k < q - 1  B_k^-1 = X^T X and the
//                   traces <K_j^-1, Sigma_k>, j > k (quirk Q1, :1039-1041);  the last workgroup to finish: expected
//                   log-likelihood (:895-990) and the ELBO (:709)
//   k_small_prior   set-up per latent GP: chol(K), chol(K)^-1, log det K, and K_j^-1 where quirk Q1 needs it (:619-622)
//
// Every reduction follows the order of the kernels it replaces (vecops.hip: k_lower_matvec, k_colops_partial / _reduce,
// k_reduce_finalize, k_dot_self, k_q1_rows / k_sum_to, k_loglike_partial / k_elbo_final), so the two paths agree to
// rounding of the compiler's contraction choices (tests/test_parity_gpu.py::test_small_path_matches_launch_schedule).
#include "gprn_internal.h"
#include "tile_mma.h"
#include "diag_tile.h"
#include "vecops.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define SMALL_MAXLD 256
// LDS: the diagonal-block kernel's buffers and the tile contraction's stages are used in turn
#define SMALL_MMA_DOUBLES (2 * 16 * (128 + 128 + 32))
#define SMALL_LDS_DOUBLES (SMALL_MMA_DOUBLES > DIAG_LDS_DOUBLES ? SMALL_MMA_DOUBLES : DIAG_LDS_DOUBLES)

__device__ __forceinline__ double sm_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the 256-thread workgroup in the order of vecops.hip's block_sum; result valid in thread 0
__device__ __forceinline__ double sm_block_sum(double v, double* sh /* 4 doubles */)
{
    v = sm_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return r;
}

// What this workgroup wrote to global memory is visible to all of ITS threads: stores acknowledged, a workgroup-scope
// fence (the waves of a workgroup share their CU's vector cache: nothing to write back or invalidate), the barrier.  Every
// use below hands data to the same workgroup.  (Until round 5 this was __threadfence(): at agent scope that writes the
// XCD's L2 back -- nothing when one evaluation's few workgroups run alone, but with 512 workgroups of a batch doing it
// three times each on an L2 full of the phase kernels' freshly written matrices k_small_tail_b took 115 us per sweep,
// 40 % of a batch: profiles/r05_batch_n45_breakdown.txt.)  Later kernels see everything at the kernel boundary.
__device__ __forceinline__ void sm_publish()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
}

// out[i] = sum_{c <= i} M[i][c] v[c] for the rows of a lower-triangular matrix of T tiles, by the workgroup's four waves:
// one wave per row and the additions in k_lower_matvec's order (lane l: columns 2 l, 2 l + 1, then + 128; then the
// shuffle tree) -- but EIGHT rows of a wave at a time: their loads go out together and their reductions interleave (one
// row after the other is a chain of an L2 round trip and six dependent shuffles per row: 26 of the 49 us of a one-tile
// half-sweep in the first version).  v in LDS or global memory; out_lds / out_g may be null.
template <int T>
__device__ __forceinline__ void small_lower_matvec(const double* M, int ld, int N, const double* v, double* out_lds, double* out_g)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NC = T;                            // 128-column chunks a row can reach into
    for (int i0 = w * 8; i0 < ld; i0 += 32) {        // rows i0 .. i0 + 7 of this wave
        double2 mv[8][NC];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                const int i = i0 + r, c = 2 * lane + 128 * cc;
                mv[r][cc] = (i < N && c <= i) ? *reinterpret_cast<const double2*>(M + (size_t)i * ld + c) : make_double2(0.0, 0.0);
            }
        double acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = i0 + r;
            acc[r] = 0.0;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                const int c = 2 * lane + 128 * cc;
                if (i < N && c <= i) {
                    acc[r] += mv[r][cc].x * v[c];
                    if (c + 1 <= i) acc[r] += mv[r][cc].y * v[c + 1];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += __shfl_down(acc[r], o, 64);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (out_lds) out_lds[i0 + r] = acc[r];
                if (out_g) out_g[i0 + r] = acc[r];
            }
        }
    }
}

// B = L L^T, X = L^-1 for T in {1, 2} tiles by one workgroup; tiles at Bm / Xm (leading dimension ld)
// N: rows that hold data -- the 16-column phases of a diagonal tile beyond them are identity padding and are not run
// (diag_tile's nph: at the reference's own N = 25 and 45 two and three of eight phases hold everything)
// ACC: a prior matrix -- panel steps by substitution (diag_tile.h)
template <int T, bool ACC = false>
__device__ __forceinline__ void small_factor(double* lds, double* Bm, double* Xm, int ld, int* info, int slot, int N)
{
    diag_tile<ACC>(lds, (gptr_t)Bm, (gptr_t)Xm, ld, info, slot, 0, T == 1 ? (N + 15) / 16 : NSB);
    if (T == 1) return;
    sm_publish();
    const size_t t10 = (size_t)GPRN_TILE * ld, t11 = t10 + GPRN_TILE;
    // L_10 = B_10 X_00^T (in place: the tile is written when all of it has been read)
    if constexpr (ACC) {
        // (a prior matrix: by substitution against L_00, 16 rows at a time per wave -- diag_tile.h trsm_rows16)
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int rb = wave; rb < NSB; rb += 4)
            trsm_rows16(lds + wave * TRSM_SCRATCH, (gptr_t)(Bm + t10) + (size_t)(16 * rb) * ld, (gcptr_t)Bm, (gcptr_t)Xm, ld);
    } else
    tile_mma<128, 128, 2, 2, 1>(lds, Bm + t10, Xm, (gptr_t)(Bm + t10), ld, 0, 0, CM_SET, GPRN_TILE, 0, 0);
    sm_publish();
    // B_11 -= L_10 L_10^T;  R_10 = -L_10 X_00 (first touch of the inverse's row)
    // (SYM: the diagonal 16 x 16 blocks accumulate from zero -- the pivots' accuracy, tile_mma.h)
    tile_mma<128, 128, 2, 2, 0, false, false, true>(lds, Bm + t10, Bm + t10, (gptr_t)(Bm + t11), ld, 0, 0, CM_SUB, GPRN_TILE, 0, 0);
    __syncthreads();
    tile_mma<128, 128, 2, 2, 0>(lds, Bm + t10, Xm, (gptr_t)(Xm + t10), ld, 0, 1, CM_SETNEG, GPRN_TILE, 0, 0);
    sm_publish();
    diag_tile<ACC>(lds, (gptr_t)(Bm + t11), (gptr_t)(Xm + t11), ld, info, slot, GPRN_TILE, (N - GPRN_TILE + 15) / 16);
    sm_publish();
    // X_10 = X_11 R_10 (in place)
    tile_mma<128, 128, 2, 2, 2>(lds, Xm + t11, Xm + t10, (gptr_t)(Xm + t10), ld, 0, 1, CM_SET, GPRN_TILE, 0, 0);
}

// the entries of B = I + D^1/2 K D^1/2 formed from K and s = sqrt(d) (LDS) on their way into the diagonal-block kernel's
// registers -- k_build_B's expression, so the tile never makes the trip to memory and back
struct DiagFromK {
    const double* K; const double* s; int ld, N;
    __device__ __forceinline__ double operator()(int row, int col) const
    {
        const bool in = row < N && col < N;
        const double kv = in ? K[(size_t)row * ld + col] : 0.0;
        const double sm = row < N ? s[row] : 0.0;
        double v = (col < N) ? sm * s[col] * kv : 0.0;
        if (row == col) v += 1.0;
        return v;
    }
};

struct SmallPhaseArgs {
    double* const* ptrs;        // [slot][GPRN_NBUF] of the phase
    const int* slot_gp;
    int N, ld, p, q;
    const double *yres, *variance;
    // The state, (p+1, q, N), in two copies: a half-sweep READS the state the sweep started from (quirk Q6, Jacobi
    // ordering: the old mu_f of the other nodes, the old mu_w -- meanfield.py:765-792, 838-865) and WRITES its rows of the new
    // one; the weight phase takes the node rows from the new one.  (In place, a workgroup that finishes early would hand
    // its new row to a neighbour that has not read the old one yet.)
    const double *mu_in, *var_in;
    double *mu_out, *var_out;
    const int* done;            // gprn_elbocalc: the stop rule has fired in an earlier sweep of the batch -- nothing to do
    double *d, *s, *pred, *z, *u, *cs, *ct;   // per-slot vectors of the phase (already offset to its first slot)
    double *trBinv, *logdetB;   // per latent GP
    int* info;
    unsigned long long* stamps; // GPRN_SMALL_STAMPS (probes): 100 MHz clock of workgroup 0 at the stages of the kernel, or null
};

// One half-sweep.  WEIGHTS: the weight phase (new mu_f, old mu_w) or the node phase; T: tiles per matrix edge.
template <bool WEIGHTS, int T>
__device__ __forceinline__ void small_phase_body(const SmallPhaseArgs& a)
{
    // (one tile: the diagonal-block kernel's 46.6 KB only -- with the vectors 59 KB, two workgroups per CU when many
    // evaluations run side by side)
    __shared__ __attribute__((aligned(16))) double lds[T == 1 ? DIAG_LDS_DOUBLES : SMALL_LDS_DOUBLES];
    __shared__ double sS[SMALL_MAXLD], sZ[SMALL_MAXLD], sD[SMALL_MAXLD], sU[SMALL_MAXLD];
    __shared__ double shs[4][64], sht[4][64], sh4[4];
    if (a.done && *a.done) return;                   // (uniform)
    const int slot = blockIdx.x, gp = a.slot_gp[slot];
    const int N = a.N, ld = a.ld, p = a.p, q = a.q, tid = threadIdx.x;
    // (a.info[slot]: only ever RAISED here, by the pivot wave of diag_tile; the host clears the rows once per call, so a
    // verdict of an early sweep of the call is still there when the host looks, as in the launch schedule)
#define SM_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    SM_STAMP(0);
    double* const Bm = a.ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = a.ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    const double* const Km = a.ptrs[(size_t)slot * GPRN_NBUF + BUF_K];
    const size_t vo = (size_t)slot * ld;
    // ---- d, s = sqrt(d), right-hand side, z = rhs / s  (k_prep_nodes / k_prep_weights; identity padding)
    for (int n = tid; n < ld; n += 256) {
        double dv = 1.0, pv = 0.0;
        if (WEIGHTS) {
            const int kk = gp - q, j = kk / p, i = kk % p;
            if (n < N) {
                const double vi = a.variance[(size_t)i * N + n];
                const double mfj = a.mu_out[(size_t)j * N + n];
                dv = (mfj * mfj + a.var_out[(size_t)j * N + n]) / vi;
                const size_t wrow = (size_t)(1 + i) * q;
                double other = 0.0;
                for (int k = 0; k < q; ++k)
                    if (k != j) other += a.mu_out[(size_t)k * N + n] * a.mu_in[(wrow + k) * N + n];
                pv = (a.yres[(size_t)i * N + n] - other) * mfj / vi;
            }
        } else {
            const int j = gp;
            if (n < N) {
                dv = 0.0;
                for (int i = 0; i < p; ++i) {
                    const double vi = a.variance[(size_t)i * N + n];
                    const size_t wrow = (size_t)(1 + i) * q;
                    const double mwj = a.mu_in[(wrow + j) * N + n];
                    const double vwj = a.var_in[(wrow + j) * N + n];
                    dv += (mwj * mwj + vwj) / vi;
                    double other = 0.0;
                    for (int k = 0; k < q; ++k)
                        if (k != j) other += a.mu_in[(wrow + k) * N + n] * a.mu_in[(size_t)k * N + n];
                    pv += (a.yres[(size_t)i * N + n] - other) * mwj / vi;
                }
            }
        }
        const double sv = sqrt(dv);
        sD[n] = dv; sS[n] = sv; sZ[n] = pv / sv;
        a.d[vo + n] = dv; a.s[vo + n] = sv; a.pred[vo + n] = pv; a.z[vo + n] = pv / sv;
    }
    __syncthreads();
    SM_STAMP(1);
    // ---- B = I + D^1/2 K D^1/2 on the lower tiles, diagonal tiles in full (k_build_B's expression); one tile: formed
    // inside the diagonal-block kernel's loads instead
    if (T > 1)
    for (int ti = 0; ti < T; ++ti)
        for (int tj = 0; tj <= ti; ++tj) {
            const int n = tj * GPRN_TILE + 2 * (tid & 63);
            const double s0 = sS[n], s1 = sS[n + 1];
            for (int it = 0; it < 32; ++it) {
                const int m = ti * GPRN_TILE + (tid >> 6) + 4 * it;
                const bool in = m < N;
                const double2 kv = in ? *reinterpret_cast<const double2*>(Km + (size_t)m * ld + n) : make_double2(0.0, 0.0);
                const double smv = in ? sS[m] : 0.0;
                double2 out;
                out.x = (n < N) ? smv * s0 * kv.x : 0.0;
                out.y = (n + 1 < N) ? smv * s1 * kv.y : 0.0;
                if (m == n) out.x += 1.0;
                if (m == n + 1) out.y += 1.0;
                *reinterpret_cast<double2*>(Bm + (size_t)m * ld + n) = out;
            }
        }
    if (T > 1) sm_publish();
    SM_STAMP(2);
    // ---- B = L L^T, X = L^-1
    if (T == 1) diag_tile_from(lds, DiagFromK{Km, sS, ld, N}, (gptr_t)Bm, (gptr_t)Xm, ld, a.info, slot, 0, (N + 15) / 16);
    else small_factor<T>(lds, Bm, Xm, ld, a.info, slot, N);
    SM_STAMP(3);
    sm_publish();
    SM_STAMP(4);
    // ---- u = X z: one wave per row, lane l takes columns 2 l, 2 l + 1 (+ 128), k_lower_matvec's order
    small_lower_matvec<T>(Xm, ld, N, sZ, sU, a.u + vo);
    __syncthreads();
    SM_STAMP(5);
    // ---- column sums over the rows of X: cs = sum x^2 (= diag B^-1), ct = sum x u (= X^T X z); per tile row the four
    // row classes in k_colops_partial's order, the tile rows added up in k_colops_reduce's
    double my_cs = 0.0, my_ct = 0.0;                 // of column `tid` (threads < ld)
    for (int c0 = 0; c0 < ld; c0 += 64) {
        const int cl = tid & 63, rl = tid >> 6;
        double acc_s = 0.0, acc_t = 0.0;             // running sums over the tile rows (thread rl == 0 holds them)
        for (int ch = c0 >> 7; ch < T; ++ch) {
            double cs = 0.0, ct = 0.0;
            {
                // the thread's 32 rows of the tile row (r = ch * 128 + rl + 4 k), loads first: one round trip to L2
                // instead of four; the additions in k_colops_partial's order
                double x[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) x[k] = Xm[(size_t)(ch * GPRN_TILE + rl + 4 * k) * ld + c0 + cl];
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    cs += x[k] * x[k];
                    ct += x[k] * sU[ch * GPRN_TILE + rl + 4 * k];
                }
            }
            __syncthreads();
            shs[rl][cl] = cs;
            sht[rl][cl] = ct;
            __syncthreads();
            if (rl == 0) {
                acc_s += (shs[0][cl] + shs[1][cl]) + (shs[2][cl] + shs[3][cl]);
                acc_t += (sht[0][cl] + sht[1][cl]) + (sht[2][cl] + sht[3][cl]);
            }
        }
        __syncthreads();
        if (rl == 0) { shs[0][cl] = acc_s; sht[0][cl] = acc_t; }
        __syncthreads();
        if (tid >= c0 && tid < c0 + 64) { my_cs = shs[0][tid - c0]; my_ct = sht[0][tid - c0]; }
        __syncthreads();
    }
    SM_STAMP(6);
    // ---- the new state of this latent GP, tr B^-1 and log det B (k_reduce_finalize: thread t owns element t)
    size_t row;
    if (gp < q) row = gp;
    else { const int kk = gp - q, j = kk / p, i = kk % p; row = (size_t)(1 + i) * q + j; }
    double tr = 0.0, ld_acc = 0.0;
    if (tid < ld) {
        a.cs[vo + tid] = my_cs;
        a.ct[vo + tid] = my_ct;
        if (tid < N) {
            a.mu_out[row * N + tid] = (sZ[tid] - my_ct) / sS[tid];
            a.var_out[row * N + tid] = (1.0 - my_cs) / sD[tid];
            tr = my_cs;
            ld_acc = log(Bm[(size_t)tid * ld + tid]);
        }
    }
    tr = sm_block_sum(tr, sh4);
    if (tid == 0) a.trBinv[gp] = tr;
    ld_acc = sm_block_sum(ld_acc, sh4);
    if (tid == 0) a.logdetB[gp] = 2.0 * ld_acc;
    SM_STAMP(7);
#undef SM_STAMP
}

struct SmallTailArgs {
    double* const* tab_node;    // [slot][GPRN_NBUF]
    double* const* tab_weight;
    const int *gp_node, *gp_weight;
    int n_node, n_weight;
    int N, ld, p, q, G;
    const double *mu, *var, *yraw, *variance;   // the NEW state of the sweep (the half-sweeps' mu_out / var_out)
    const double* s_node;       // s = sqrt(d) of the node slots
    double* const* Kinv;        // [q] device pointers: K_j^-1 (lower), null where not held
    double* scratch;            // [n_node + n_weight][ld]
    const double* logdetK;
    double* scal;               // logdetB | trBinv | muKmu | q1 [q * q]
    double* out4;
    unsigned* ticket;
    // gprn_elbocalc (or null): the loop of meanfield.py:626-649 on the device.  ctl: [0] done, [1] iterNumber, [2] converged;
    // hist: the batch's ELBO values, this sweep's at hist[hist_at]; last3: the three latest values of the loop
    int* ctl;                   // [0] done, [1] iterNumber, [2] converged, [3] the last sweep that ran
    double *hist, *last3;
    int sweep, hist_at, max_iter;
    const int* info;            // the pivot verdicts of this evaluation (set-up, node phase, weight phase), n_info words
    int n_info;
};

// log-likelihood terms of k_loglike_partial for block 0 (with N <= 256 the other 31 blocks of that kernel are empty
// and contribute exact zeros); result valid in thread 0
__device__ __forceinline__ void small_loglike(const SmallTailArgs& a, double* sh4, double& t1, double& t2, double& t3)
{
    const double TWO_PI = 6.283185307179586;
    t1 = t2 = t3 = 0.0;
    const int n = threadIdx.x;
    if (n < a.N) {
        for (int i = 0; i < a.p; ++i) {
            const double vi = a.variance[(size_t)i * a.N + n];
            const size_t wrow = (size_t)(1 + i) * a.q;
            t1 += log(TWO_PI * vi);
            double fit = 0.0, cross = 0.0;
            for (int j = 0; j < a.q; ++j) {
                const double mf = a.mu[(size_t)j * a.N + n], vf = a.var[(size_t)j * a.N + n];
                const double mw = a.mu[(wrow + j) * a.N + n], vw = a.var[(wrow + j) * a.N + n];
                fit += mw * mf;
                cross += vf * (mw * mw) + vw * (mf * mf) + vf * vw;
            }
            const double resid = a.yraw[(size_t)i * a.N + n] - fit;
            t2 += resid * resid / vi;
            t3 += cross / vi;
        }
    }
    t1 = sm_block_sum(t1, sh4);
    t2 = sm_block_sum(t2, sh4);
    t3 = sm_block_sum(t3, sh4);
}

template <int T>
__device__ __forceinline__ void small_tail_body(const SmallTailArgs& a)
{
    __shared__ __attribute__((aligned(16))) double lds[SMALL_MMA_DOUBLES];
    __shared__ double sh4[4];
    __shared__ unsigned last;
    if (a.ctl && a.ctl[0]) return;                   // (uniform)
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_node = wg < a.n_node;
    const int slot = is_node ? wg : wg - a.n_node;
    double* const* row = (is_node ? a.tab_node : a.tab_weight) + (size_t)slot * GPRN_NBUF;
    const int gp = is_node ? a.gp_node[slot] : a.gp_weight[slot];
    const int N = a.N, ld = a.ld, G = a.G, q = a.q;
    double* const tmp = a.scratch + (size_t)wg * ld;
    double* const muKmu = a.scal + 2 * (size_t)G;
    double* const q1 = a.scal + 3 * (size_t)G;
    // ---- mu^T K^-1 mu = |L_K^-1 m|^2, m = row gp of the state as it lies in memory (quirk Q2 for the weights)
    {
        small_lower_matvec<T>(row[BUF_KLINV], ld, N, a.mu + (size_t)gp * N, nullptr, tmp);
        sm_publish();
        double acc = 0.0;
        for (int n = tid; n < N; n += 256) { const double x = tmp[n]; acc += x * x; }
        acc = sm_block_sum(acc, sh4);
        if (tid == 0) muKmu[gp] = acc;
        __syncthreads();
    }
    // ---- quirk Q1: node k < q - 1 forms lower(B_k^-1) = lower(X^T X) in its B buffer (L is not needed any more) and the
    // traces <K_j^-1, Sigma_k>, j > k (k_q1_rows, k_sum_to)
    if (is_node && gp < q - 1) {
        double* const Bm = row[BUF_B];
        const double* const Xm = row[BUF_X];
        for (int ta = T - 1; ta >= 0; --ta)
            for (int tb = 0; tb <= ta; ++tb) {
                const size_t oa = (size_t)ta * GPRN_TILE * ld + (size_t)ta * GPRN_TILE, ob = (size_t)ta * GPRN_TILE * ld + (size_t)tb * GPRN_TILE;
                __syncthreads();
                tile_mma<128, 128, 2, 2, 0>(lds, Xm + oa, Xm + ob, (gptr_t)(Bm + ob), ld, 1, 1, CM_SET, ld - ta * GPRN_TILE, 0, 0);
            }
        sm_publish();
        const double* sv = a.s_node + (size_t)slot * ld;
        for (int j = gp + 1; j < q; ++j) {
            const double* Kj = a.Kinv[j];
            for (int m = w; m < N; m += 4) {
                const double* kr = Kj + (size_t)m * ld;
                const double* br = Bm + (size_t)m * ld;
                double acc = 0.0;
                for (int n = lane; n < m; n += 64) acc -= kr[n] * br[n] / sv[n];
                acc = sm_wave_sum(acc);
                if (lane == 0) {
                    const double smv = sv[m];
                    tmp[m] = 2.0 * acc / smv + kr[m] * (1.0 - br[m]) / (smv * smv);
                }
            }
            sm_publish();
            double acc = 0.0;
            for (int i = tid; i < N; i += 256) acc += tmp[i];
            acc = sm_block_sum(acc, sh4);
            if (tid == 0) q1[(size_t)j * q + gp] = acc;
            __syncthreads();
        }
    }
    // ---- the last workgroup to get here assembles the ELBO (k_loglike_partial + k_elbo_final).  What it reads of the
    // others -- mu^T K^-1 mu and the Q1 traces -- thread 0 has stored: its release (agent scope: the reader may sit on
    // another XCD) orders them before the ticket; the reader's loads of them are agent-scope atomics
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last = atomicAdd(a.ticket, 1u) + 1 == gridDim.x ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;                              // (uniform)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double t1, t2, t3;
    small_loglike(a, sh4, t1, t2, t3);
    if (tid == 0) {
        atomicExch(a.ticket, 0u);
        const double TWO_PI = 6.283185307179586;
        auto sc_at = [&](size_t i) { return __hip_atomic_load(a.scal + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        const double* logdetK = a.logdetK;
        const int p = a.p;
        const double logl = -0.5 * t1 - 0.5 * t2 - 0.5 * t3;
        double ent = 0.0, logp = 0.0;
        for (int g = 0; g < G; ++g) {
            ent += 0.5 * (logdetK[g] - sc_at(g));
            double tr = sc_at(G + g);
            if (g < q)
                for (int k = 0; k < g; ++k) tr += sc_at(3 * (size_t)G + g * q + k);   // cumulative sumSigmaF, quirk Q1
            logp += -0.5 * logdetK[g] - 0.5 * (sc_at(2 * (size_t)G + g) + tr);
        }
        const double cst = (double)q * (p + 1) * N;
        ent += 0.5 * cst * (1.0 + log(TWO_PI));
        logp += -0.5 * cst * log(TWO_PI);
        const double elbo = (logl + logp + ent) / q;
        a.out4[0] = elbo;
        a.out4[1] = logl;
        a.out4[2] = logp;
        a.out4[3] = ent;
        if (a.ctl) {
            // The loop of ELBOcalc (meanfield.py:626-649): sweep 0 is the discarded one (quirk Q7: its ELBO is kept, its
            // update is not); sweep s >= 1 is loop trip s.  Stop rule :640-643 on the last three values: NumPy's
            // np.std / np.mean of three numbers, operation by operation (no contraction: every product and sum rounds)
            a.hist[a.hist_at] = elbo;
            a.ctl[3] = a.sweep;
            // a matrix that was not positive definite, or a state that has left the finite numbers: NaN stays NaN (jax's
            // cholesky semantics, meanfield.py:71-89), the stop rule never fires on it and the reference's loop runs to
            // max_iter to return it -- so does this one, without the sweeps
            bool failed = !(elbo == elbo);
            for (int i = 0; i < a.n_info && !failed; ++i) failed = a.info[i] > 0;
            if (failed) {
                a.ctl[0] = 1; a.ctl[1] = a.max_iter; a.ctl[2] = 0;
            } else if (a.sweep >= 1) {
                // (quirk Q7: elboArray[0], the discarded first call's value, IS trip 1's -- same input, same arithmetic; the
                // host enqueues sweep 0 only for max_iter = 0)
                const double e0 = a.last3[1], e1 = a.sweep == 1 ? elbo : a.last3[2], e2 = elbo;
                a.last3[0] = e0; a.last3[1] = e1; a.last3[2] = e2;
                a.ctl[1] = a.sweep;
                bool stop = false;
                if (a.sweep > 3) {
                    const double mean = __ddiv_rn(__dadd_rn(__dadd_rn(e0, e1), e2), 3.0);
                    const double d0 = __dsub_rn(e0, mean), d1 = __dsub_rn(e1, mean), d2 = __dsub_rn(e2, mean);
                    const double var3 = __ddiv_rn(__dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2)), 3.0);
                    const double crit = fabs(__ddiv_rn(__dsqrt_rn(var3), mean));
                    if (crit < 1e-3 && crit != 0.0) { stop = true; a.ctl[2] = 1; }
                }
                if (stop || a.sweep >= a.max_iter) a.ctl[0] = 1;
            } else {
                a.last3[2] = elbo;                   // elboArray[0]; trip 1 shifts it down
            }
        }
    }
}

struct SmallPriorArgs {
    double* const* ptrs;        // [job][GPRN_NBUF]: BUF_B scratch, BUF_X = chol(K)^-1 out, BUF_K = K
    const int* job_gp;
    double* const* Kinv_out;    // [job]: where lower(K^-1) goes, or null
    int N, ld;
    double* logdetK;            // per latent GP
    int* info;
};

// chol(K) and its inverse for one latent GP per workgroup; log det K; K^-1 where asked
template <int T, bool ACC>
__device__ __forceinline__ void small_prior_body(const SmallPriorArgs& a)
{
    __shared__ __attribute__((aligned(16))) double lds[SMALL_LDS_DOUBLES];
    __shared__ double sh4[4];
    const int job = blockIdx.x, tid = threadIdx.x, ld = a.ld, N = a.N;
    double* const Bm = a.ptrs[(size_t)job * GPRN_NBUF + BUF_B];
    double* const Xm = a.ptrs[(size_t)job * GPRN_NBUF + BUF_X];
    const double* const Km = a.ptrs[(size_t)job * GPRN_NBUF + BUF_K];
    if (Bm != Km) {                                  // (a.info[job]: raised here, cleared by the host before the launch)
        for (int i = tid * 2; i < ld * ld; i += 512)
            *reinterpret_cast<double2*>(Bm + i) = *reinterpret_cast<const double2*>(Km + i);
        sm_publish();
    }
    small_factor<T, ACC>(lds, Bm, Xm, ld, a.info, job, N);
    sm_publish();
    double acc = 0.0;
    for (int n = tid; n < N; n += 256) acc += log(Bm[(size_t)n * ld + n]);
    acc = sm_block_sum(acc, sh4);
    if (tid == 0) a.logdetK[a.job_gp[job]] = 2.0 * acc;
    double* const Ki = a.Kinv_out[job];
    if (Ki) {                                        // (uniform)
        for (int ta = T - 1; ta >= 0; --ta)
            for (int tb = 0; tb <= ta; ++tb) {
                const size_t oa = (size_t)ta * GPRN_TILE * ld + (size_t)ta * GPRN_TILE, ob = (size_t)ta * GPRN_TILE * ld + (size_t)tb * GPRN_TILE;
                __syncthreads();
                tile_mma<128, 128, 2, 2, 0>(lds, Xm + oa, Xm + ob, (gptr_t)(Ki + ob), ld, 1, 1, CM_SET, ld - ta * GPRN_TILE, 0, 0);
            }
    }
}

// ---- the kernels: one problem per launch (grid x = latent GP), or -- gprn_elbocalc_batch -- many independent evaluations
// of the same problem shape per launch (grid y = evaluation, its arguments in device memory)
template <bool WEIGHTS, int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_small_phase(SmallPhaseArgs a) { small_phase_body<WEIGHTS, T>(a); }
// (one tile, many evaluations: up to two workgroups per CU -- 222 registers per lane fit twice)
template <bool WEIGHTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void k_small_phase_b(const SmallPhaseArgs* __restrict__ lanes) { small_phase_body<WEIGHTS, 1>(lanes[blockIdx.y]); }

template <int T>
__global__ __launch_bounds__(256)
void k_small_tail(SmallTailArgs a) { small_tail_body<T>(a); }
template <int T>
__global__ __launch_bounds__(256)
void k_small_tail_b(const SmallTailArgs* __restrict__ lanes, int sweep, int hist_at, int max_iter)
{
    SmallTailArgs a = lanes[blockIdx.y];
    a.sweep = sweep; a.hist_at = hist_at; a.max_iter = max_iter;
    small_tail_body<T>(a);
}

template <int T, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_small_prior(SmallPriorArgs a) { small_prior_body<T, ACC>(a); }
template <int T, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_small_prior_b(const SmallPriorArgs* __restrict__ lanes) { small_prior_body<T, ACC>(lanes[blockIdx.y]); }

// ------------------------------------------------------------------ launchers
// One tile (N <= 128) by default.  Two tiles work too (option "small_path" = 2; same tests) but do not pay: the four
// 128^3 products of the 2 x 2 blocked form on ONE workgroup's four waves make a half-sweep 170 us, where the launch
// schedule spreads them over the device (N = 200, p = q = 1: 2.14 ms per nELBO evaluation against 1.88).
bool small_applies(const gprn_ctx* c)
{
    const int max_T = c->small_opt == 2 ? 2 : 1;
    return c->small_opt != 0 && c->T >= 1 && c->T <= max_T && !c->comm && !c->shm && !c->keep_sigma && c->world == 1;
}

int small_phase(gprn_ctx* c, bool weights, const int* d_slot_gp, int nslots, const double* mu_in, const double* var_in,
                double* mu_out, double* var_out, const int* done)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_DIAG);
    const size_t o = (size_t)c->slot0 * c->ld;
    SmallPhaseArgs a{(double* const*)c->d_ptrs, d_slot_gp, c->N, c->ld, c->p, c->q, c->d_yres, c->d_variance,
                     mu_in, var_in, mu_out, var_out, done,
                     c->d_d + o, c->d_s + o, c->d_pred + o, c->d_z + o, c->d_u + o, c->d_cs + o, c->d_ct + o,
                     c->d_trBinv, c->d_logdetB, c->d_info_cur, weights ? nullptr : c->d_small_stamps};
#define GO(W, TT) hipLaunchKernelGGL((k_small_phase<W, TT>), dim3(nslots), dim3(256), 0, c->stream, a)
    if (c->T == 1) { if (weights) GO(true, 1); else GO(false, 1); }
    else { if (weights) GO(true, 2); else GO(false, 2); }
#undef GO
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

int small_tail(gprn_ctx* c, double* out4, double* scal, const double* mu, const double* var, const SmallLoop* loop)
{
    const int nn = (int)c->loc_nodes.size(), nw = (int)c->loc_weights.size();
    if (nn + nw == 0) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    SmallTailArgs a{(double* const*)c->tab_node, (double* const*)c->tab_weight, c->d_slotgp_node, c->d_slotgp_weight, nn, nw,
                    c->N, c->ld, c->p, c->q, c->G, mu, var, c->d_yraw, c->d_variance, c->d_s,
                    (double* const*)c->d_kinv_tab, c->d_u, c->d_logdetK, scal, out4, c->d_small_ticket,
                    loop ? loop->ctl : nullptr, loop ? loop->hist : nullptr, loop ? loop->last3 : nullptr,
                    loop ? loop->sweep : 0, loop ? loop->hist_at : 0, loop ? loop->max_iter : 0,
                    c->d_info, 3 * c->nslot};
    if (c->T == 1) hipLaunchKernelGGL(k_small_tail<1>, dim3(nn + nw), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(k_small_tail<2>, dim3(nn + nw), dim3(256), 0, c->stream, a);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

int small_prior(gprn_ctx* c, double** d_tab, const int* d_job_gp, double** d_kinv_out, int njobs, int* d_info)
{
    if (!njobs) return GPRN_OK;
    prof_begin(c, GPRN_T_DIAG);
    SmallPriorArgs a{(double* const*)d_tab, d_job_gp, (double* const*)d_kinv_out, c->N, c->ld, c->d_logdetK, d_info};
    // (prior matrices: panel steps by substitution unless option "accurate_factor" = 0)
    const bool acc = c->acc_opt != 0;
#define GO_SP(TT, ACC) hipLaunchKernelGGL((k_small_prior<TT, ACC>), dim3(njobs), dim3(256), 0, c->stream, a)
    if (c->T == 1) { if (acc) GO_SP(1, true); else GO_SP(1, false); }
    else { if (acc) GO_SP(2, true); else GO_SP(2, false); }
#undef GO_SP
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ------------------------------------------------------------------ many independent evaluations in one go
// gprn_elbocalc_batch: B evaluations of inference.nELBO at B parameter vectors -- what an optimiser's population or
// emcee's walkers ask for one by one (meanfield.py:1095-1111, 1222-1260) -- as ONE stream of launches: every launch
// covers all B problems (grid y), each with its own matrices, state, loop control and stop rule; an evaluation that has
// converged turns its workgroups into no-ops while the others go on.  A one-tile problem keeps G of the device's 256
// CUs busy; B of them side by side fill it.
struct SmallBatchMem {
    int cap = 0;                      // evaluations the buffers hold
    int G = 0, q = 0, p = 0, N = 0, ld = 0;
    double *mats = nullptr;           // [cap][4 G + q][ld * ld]: K, KLinv, wsB, wsX per latent GP, K_j^-1 per node
    double *vecs = nullptr;           // [cap][7][G * ld]
    double *state = nullptr;          // [4][cap][d]: mu A, var A, mu B, var B
    double *yv = nullptr;             // [2][cap][p N]: y - mean, variance
    double *scal = nullptr, *logdetK = nullptr, *out4 = nullptr, *hist = nullptr;
    int *ctl = nullptr, *info = nullptr, *gp_ids = nullptr;
    unsigned* ticket = nullptr;
    double** ptrs = nullptr;          // pointer tables: [cap] x (3 tables of G x 4, kinv_tab q, kinv_out G, K pointers G)
    double** kptr_dense = nullptr;    // [cap][G]: where the fill puts evaluation b's matrix g
    void* programs = nullptr;         // [cap][G] fill programs
    SmallPhaseArgs* phase_args = nullptr;   // [2 parity][2 phase][cap]
    SmallTailArgs* tail_args = nullptr;     // [2 parity][cap]
    SmallPriorArgs* prior_args = nullptr;   // [cap]
    char *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
};
#define SB_K 8                         // sweeps per batch of launches (one synchronisation each)

void small_batch_free(gprn_ctx* c)
{
    SmallBatchMem* m = (SmallBatchMem*)c->small_batch;
    if (!m) return;
    void* dev[] = {m->mats, m->vecs, m->state, m->yv, m->scal, m->logdetK, m->out4, m->hist, m->ctl, m->info, m->gp_ids,
                   m->ticket, m->ptrs, m->kptr_dense, m->programs, m->phase_args, m->tail_args, m->prior_args};
    for (void* ptr : dev) if (ptr) hipFree(ptr);
    if (m->pin_in) hipHostFree(m->pin_in);
    if (m->pin_out) hipHostFree(m->pin_out);
    delete m;
    c->small_batch = nullptr;
}

template <typename TT>
static int sb_alloc(gprn_ctx* c, TT** ptr, size_t count)
{
    *ptr = nullptr;
    if (hipMalloc((void**)ptr, std::max<size_t>(count, 1) * sizeof(TT)) != hipSuccess) { c->err = "hipMalloc (evaluation batch)"; return GPRN_E_NOMEM; }
    return GPRN_OK;
}
#define SB_TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)

// evaluations one chunk may hold: what the memory budget pays for (4 G + q matrices of ld^2 doubles each and small change
// per evaluation), 16 at least -- an emcee run with thousands of walkers is split, not refused
int small_batch_chunk(gprn_ctx* c)
{
    const size_t nn = (size_t)c->ld * c->ld, d = (size_t)(c->p + 1) * c->q * c->N;
    const size_t per = ((4 * (size_t)c->G + c->q) * nn + 7 * (size_t)c->G * c->ld + 6 * d + 4 * (size_t)c->p * c->N + 256) * sizeof(double) +
                       (size_t)c->G * fill_program_bytes() * 2;
    return (int)std::max<size_t>(16, std::min<size_t>(batch_budget_bytes(c) / per, 32768));   // (grid y = evaluations)
}

static int small_batch_ensure(gprn_ctx* c, int n_eval)
{
    SmallBatchMem* m = (SmallBatchMem*)c->small_batch;
    const int G = c->G, q = c->q, p = c->p, N = c->N, ld = c->ld;
    if (m && m->cap >= n_eval && m->G == G && m->q == q && m->p == p && m->N == N && m->ld == ld) return GPRN_OK;
    small_batch_free(c);
    m = new SmallBatchMem();
    c->small_batch = m;
    const int cap = std::max(n_eval, 16);
    m->G = G; m->q = q; m->p = p; m->N = N; m->ld = ld;
    const size_t nn = (size_t)ld * ld, d = (size_t)(p + 1) * q * N, pn = (size_t)p * N, nscal = 3 * (size_t)G + (size_t)q * q;
    const size_t nmat = 4 * (size_t)G + q, nptr = 3 * (size_t)G * GPRN_NBUF + q + 2 * (size_t)G;
    SB_TRY(sb_alloc(c, &m->mats, (size_t)cap * nmat * nn));
    SB_TRY(sb_alloc(c, &m->vecs, (size_t)cap * 7 * G * ld));
    SB_TRY(sb_alloc(c, &m->state, 4 * (size_t)cap * d));
    SB_TRY(sb_alloc(c, &m->yv, 2 * (size_t)cap * pn));
    SB_TRY(sb_alloc(c, &m->scal, (size_t)cap * nscal));
    SB_TRY(sb_alloc(c, &m->logdetK, (size_t)cap * G));
    SB_TRY(sb_alloc(c, &m->out4, (size_t)cap * 4));
    SB_TRY(sb_alloc(c, &m->hist, (size_t)cap * (SB_K + 4)));
    SB_TRY(sb_alloc(c, &m->ctl, (size_t)cap * 4));
    SB_TRY(sb_alloc(c, &m->info, (size_t)cap * 3 * G));
    SB_TRY(sb_alloc(c, &m->gp_ids, (size_t)G));
    SB_TRY(sb_alloc(c, &m->ticket, (size_t)cap));
    SB_TRY(sb_alloc(c, &m->ptrs, (size_t)cap * nptr));
    SB_TRY(sb_alloc(c, &m->kptr_dense, (size_t)cap * G));
    if (hipMalloc(&m->programs, (size_t)cap * G * fill_program_bytes()) != hipSuccess) { c->err = "hipMalloc (fill programs)"; return GPRN_E_NOMEM; }
    SB_TRY(sb_alloc(c, &m->phase_args, 4 * (size_t)cap));
    SB_TRY(sb_alloc(c, &m->tail_args, 2 * (size_t)cap));
    SB_TRY(sb_alloc(c, &m->prior_args, (size_t)cap));
    HIP_TRY(c, hipMemset(m->ticket, 0, (size_t)cap * sizeof(unsigned)));
    HIP_TRY(c, hipMemset(m->info, 0, (size_t)cap * 3 * G * sizeof(int)));      // (the kernels write the entries they use)
    HIP_TRY(c, hipMemset(m->scal, 0, (size_t)cap * nscal * sizeof(double)));
    std::vector<int> ids(G);
    for (int g = 0; g < G; ++g) ids[g] = g;
    HIP_TRY(c, hipMemcpy(m->gp_ids, ids.data(), G * sizeof(int), hipMemcpyHostToDevice));
    // ---- the fixed part of every evaluation's arguments: pointer tables and argument blocks
    std::vector<double*> hp((size_t)cap * nptr, nullptr);
    std::vector<SmallPhaseArgs> pa(4 * (size_t)cap);
    std::vector<SmallTailArgs> ta(2 * (size_t)cap);
    std::vector<SmallPriorArgs> pr((size_t)cap);
    for (int b = 0; b < cap; ++b) {
        double* const mb = m->mats + (size_t)b * nmat * nn;
        auto Kp = [&](int g) { return mb + (size_t)g * nn; };
        auto KLp = [&](int g) { return mb + ((size_t)G + g) * nn; };
        auto Bp = [&](int g) { return mb + (2 * (size_t)G + g) * nn; };
        auto Xp = [&](int g) { return mb + (3 * (size_t)G + g) * nn; };
        auto Kinvp = [&](int j) { return mb + (4 * (size_t)G + j) * nn; };
        double** const hb = hp.data() + (size_t)b * nptr;           // host image of this evaluation's tables
        double** const db = m->ptrs + (size_t)b * nptr;             // ... and where it lies on the device
        double** const h_setup = hb; double** const h_node = hb + (size_t)G * GPRN_NBUF; double** const h_weight = hb + 2 * (size_t)G * GPRN_NBUF;
        double** const h_kinv_tab = hb + 3 * (size_t)G * GPRN_NBUF; double** const h_kinv_out = h_kinv_tab + q; double** const h_Kptr = h_kinv_out + G;
        for (int g = 0; g < G; ++g) {
            double* row[GPRN_NBUF] = {Bp(g), KLp(g), Kp(g), KLp(g)};      // set-up: BUF_B scratch, BUF_X = chol(K)^-1, BUF_K
            for (int k = 0; k < GPRN_NBUF; ++k) h_setup[(size_t)g * GPRN_NBUF + k] = row[k];
            double* srow[GPRN_NBUF] = {Bp(g), Xp(g), Kp(g), KLp(g)};       // sweeps: B, X, K, chol(K)^-1
            double** const dst = g < q ? h_node + (size_t)g * GPRN_NBUF : h_weight + (size_t)(g - q) * GPRN_NBUF;
            for (int k = 0; k < GPRN_NBUF; ++k) dst[k] = srow[k];
            h_kinv_out[g] = (g >= 1 && g < q) ? Kinvp(g) : nullptr;
            h_Kptr[g] = Kp(g);
        }
        for (int j = 0; j < q; ++j) h_kinv_tab[j] = j >= 1 ? Kinvp(j) : nullptr;
        double** const d_setup = db; double** const d_node = db + (size_t)G * GPRN_NBUF; double** const d_weight = db + 2 * (size_t)G * GPRN_NBUF;
        double** const d_kinv_tab = db + 3 * (size_t)G * GPRN_NBUF; double** const d_kinv_out = d_kinv_tab + q;
        double* const vb = m->vecs + (size_t)b * 7 * G * ld;
        auto vec = [&](int which, int slot0) { return vb + ((size_t)which * G + slot0) * ld; };
        double* const yres = m->yv + (size_t)b * pn; double* const variance = m->yv + ((size_t)cap + b) * pn;
        double* const st[4] = {m->state + (size_t)b * d, m->state + ((size_t)cap + b) * d, m->state + (2 * (size_t)cap + b) * d,
                               m->state + (3 * (size_t)cap + b) * d};
        double* const scal = m->scal + (size_t)b * nscal;
        int* const ctl = m->ctl + (size_t)b * 4;
        int* const info = m->info + (size_t)b * 3 * G;
        for (int par = 0; par < 2; ++par) {
            const double* mu_in = par ? st[2] : st[0]; const double* var_in = par ? st[3] : st[1];
            double* mu_out = par ? st[0] : st[2]; double* var_out = par ? st[1] : st[3];
            for (int ph = 0; ph < 2; ++ph) {
                const int s0 = ph ? q : 0;
                pa[((size_t)par * 2 + ph) * cap + b] = SmallPhaseArgs{
                    (double* const*)(ph ? d_weight : d_node), m->gp_ids + s0, N, ld, p, q, yres, variance,
                    mu_in, var_in, mu_out, var_out, ctl,
                    vec(0, s0), vec(1, s0), vec(2, s0), vec(3, s0), vec(4, s0), vec(5, s0), vec(6, s0),
                    scal + G, scal, info + (size_t)(1 + ph) * G, nullptr};
            }
            ta[(size_t)par * cap + b] = SmallTailArgs{
                (double* const*)d_node, (double* const*)d_weight, m->gp_ids, m->gp_ids + q, q, G - q, N, ld, p, q, G,
                mu_out, var_out, c->d_yraw, variance, vec(1, 0), (double* const*)d_kinv_tab, vec(4, 0), m->logdetK + (size_t)b * G,
                scal, m->out4 + (size_t)b * 4, m->ticket + b, ctl, m->hist + (size_t)b * (SB_K + 4),
                m->hist + (size_t)b * (SB_K + 4) + SB_K, 0, 0, 0, info, 3 * G};
        }
        pr[b] = SmallPriorArgs{(double* const*)d_setup, m->gp_ids, (double* const*)d_kinv_out, N, ld, m->logdetK + (size_t)b * G, info};
    }
    HIP_TRY(c, hipMemcpy(m->ptrs, hp.data(), hp.size() * sizeof(double*), hipMemcpyHostToDevice));
    {
        std::vector<double*> kd((size_t)cap * G);
        for (int b = 0; b < cap; ++b)
            for (int g = 0; g < G; ++g) kd[(size_t)b * G + g] = m->mats + ((size_t)b * nmat + g) * nn;
        HIP_TRY(c, hipMemcpy(m->kptr_dense, kd.data(), kd.size() * sizeof(double*), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMemcpy(m->phase_args, pa.data(), pa.size() * sizeof(SmallPhaseArgs), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(m->tail_args, ta.data(), ta.size() * sizeof(SmallTailArgs), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(m->prior_args, pr.data(), pr.size() * sizeof(SmallPriorArgs), hipMemcpyHostToDevice));
    // pinned staging: in = programs | y - mean | variance | mu | var;  out = ctl | hist | info | the four state copies
    m->pin_in_bytes = (size_t)cap * G * fill_program_bytes() + (2 * (size_t)cap * pn + 2 * (size_t)cap * d) * sizeof(double);
    m->pin_out_bytes = (size_t)cap * (4 * sizeof(int) + (SB_K + 4) * sizeof(double) + 3 * G * sizeof(int)) + 4 * (size_t)cap * d * sizeof(double) + 64;
    HIP_TRY(c, hipHostMalloc((void**)&m->pin_in, m->pin_in_bytes, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&m->pin_out, m->pin_out_bytes, hipHostMallocDefault));
    m->cap = cap;
    return GPRN_OK;
}

int small_batch_elbocalc(gprn_ctx* c, int n_eval, const double* kparams, int n_kpar, const double* y_resid, const double* jitters,
                         const double* mu, const double* var, int max_iter, double* elbo, int* iters, int* conv, int* info,
                         double* mu_out, double* var_out)
{
    if (c->T != 1 || !small_applies(c)) { c->err = "elbocalc_batch: one-tile problems on one rank only"; return GPRN_E_UNSUPPORTED; }
    const int G = c->G, p = c->p, N = c->N, B = n_eval;
    int kp_total = 0;
    for (int g = 0; g < G; ++g) {
        if (!c->kspec[g].set || c->kspec[g].uploaded) { c->err = "elbocalc_batch: every latent GP needs a device kernel program"; return GPRN_E_UNSUPPORTED; }
        kp_total += c->kspec[g].n_params;
    }
    if (kp_total != n_kpar) { c->err = "elbocalc_batch: kernel_params has the wrong length per evaluation"; return GPRN_E_ARG; }
    // GPRN_BATCH_TIMERS=1 (probes): where the host's time of a call goes -- staging, enqueue, waits, read-back -- on stderr
    static int timers_env = -1;
    if (timers_env < 0) { const char* e = getenv("GPRN_BATCH_TIMERS"); timers_env = e ? atoi(e) : 0; }
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
    double us_stage = 0.0, us_enqueue = 0.0, us_wait = 0.0, us_host = 0.0;
    SB_TRY(small_batch_ensure(c, B));
    SmallBatchMem* m = (SmallBatchMem*)c->small_batch;
    const int cap = m->cap;
    const double us_ensure = since(t_begin);
    auto t_mark = std::chrono::steady_clock::now();
    const size_t d = (size_t)(p + 1) * c->q * N, pn = (size_t)p * N, pb = fill_program_bytes();
    // ---- inputs through the pinned buffer
    char* const pg_h = m->pin_in;
    double* const yres_h = (double*)(pg_h + (size_t)cap * G * pb);
    double* const var_h = yres_h + (size_t)cap * pn;
    double* const mu0_h = var_h + (size_t)cap * pn;
    double* const v0_h = mu0_h + (size_t)cap * d;
    for (int b = 0; b < B; ++b) {
        const double* kp = kparams + (size_t)b * n_kpar;
        for (int g = 0; g < G; ++g) {
            if (!fill_program_with(c->kspec[g], kp, pg_h + ((size_t)b * G + g) * pb)) {
                c->err = "elbocalc_batch: a kernel that is not an even function of t_i - t_j"; return GPRN_E_UNSUPPORTED;
            }
            kp += c->kspec[g].n_params;
        }
        memcpy(yres_h + (size_t)b * pn, y_resid + (size_t)b * pn, pn * sizeof(double));
        for (int i = 0; i < p; ++i) {
            const double j2 = jitters[(size_t)b * p + i] * jitters[(size_t)b * p + i];
            for (int n = 0; n < N; ++n) var_h[(size_t)b * pn + (size_t)i * N + n] = j2 + c->h_yerr2[(size_t)i * N + n];
        }
        memcpy(mu0_h + (size_t)b * d, mu + (size_t)b * d, d * sizeof(double));
        memcpy(v0_h + (size_t)b * d, var + (size_t)b * d, d * sizeof(double));
    }
    hipStream_t st = c->stream;
    us_stage = since(t_mark); t_mark = std::chrono::steady_clock::now();
    HIP_TRY(c, hipMemcpyAsync(m->programs, pg_h, (size_t)B * G * pb, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(m->yv, yres_h, (size_t)B * pn * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(m->yv + (size_t)cap * pn, var_h, (size_t)B * pn * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(m->state, mu0_h, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(m->state + (size_t)cap * d, v0_h, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(m->ctl, 0, (size_t)B * 4 * sizeof(int), st));
    HIP_TRY(c, hipMemsetAsync(m->info, 0, (size_t)B * 3 * G * sizeof(int), st));    // (the kernels only raise them)
    // ---- set-up: every evaluation's G covariance matrices in one launch, their factors in another
    SB_TRY(launch_fill_batch(c, m->programs, (double* const*)m->kptr_dense, B * G));
    prof_begin(c, GPRN_T_DIAG);
    if (c->acc_opt != 0) hipLaunchKernelGGL((k_small_prior_b<1, true>), dim3(G, B), dim3(256), 0, st, (const SmallPriorArgs*)m->prior_args);
    else hipLaunchKernelGGL((k_small_prior_b<1, false>), dim3(G, B), dim3(256), 0, st, (const SmallPriorArgs*)m->prior_args);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    // ---- the loop, SB_K sweeps per synchronisation
    char* const po = m->pin_out;
    int* const ctl_h = (int*)po;
    double* const hist_h = (double*)(po + (size_t)cap * 4 * sizeof(int));
    int* const info_h = (int*)(hist_h + (size_t)cap * (SB_K + 4));
    double* const st_h = (double*)(((uintptr_t)(info_h + (size_t)cap * 3 * G) + 63) & ~(uintptr_t)63);
    std::vector<char> was_done(B, 0);
    for (int b = 0; b < B; ++b) { elbo[b] = 0.0; iters[b] = 0; conv[b] = 0; info[b] = 0; }
    // (quirk Q7: sweep 0 -- the first ELBOaux call, whose update is discarded -- and trip 1 are the same computation on the
    // same input; it runs once, as trip 1.  Only max_iter = 0 enqueues sweep 0.)
    int s = max_iter >= 1 ? 1 : 0;
    bool all_done = false;
    const int q = c->q;
    while (!all_done && s <= max_iter) {
        const int s0 = s;
        int nb = 0;
        const int nb_max = s0 <= 1 ? 4 : SB_K;                 // (no verdict before trip 4, and most warm starts stop there)
        for (; nb < nb_max && s <= max_iter; ++nb, ++s) {
            const int par = (s <= 1 || (s & 1)) ? 0 : 1;          // sweep 0 and trip 1 start from copy A, then they alternate
            prof_begin(c, GPRN_T_DIAG);
            hipLaunchKernelGGL((k_small_phase_b<false>), dim3(q, B), dim3(256), 0, st, (const SmallPhaseArgs*)(m->phase_args + ((size_t)par * 2 + 0) * cap));
            hipLaunchKernelGGL((k_small_phase_b<true>), dim3(G - q, B), dim3(256), 0, st, (const SmallPhaseArgs*)(m->phase_args + ((size_t)par * 2 + 1) * cap));
            prof_end(c);
            prof_begin(c, GPRN_T_VEC);
            hipLaunchKernelGGL(k_small_tail_b<1>, dim3(G, B), dim3(256), 0, st, (const SmallTailArgs*)(m->tail_args + (size_t)par * cap), s, nb, max_iter);
            prof_end(c);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(ctl_h, m->ctl, (size_t)B * 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(hist_h, m->hist, (size_t)B * (SB_K + 4) * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(info_h, m->info, (size_t)B * 3 * G * sizeof(int), hipMemcpyDeviceToHost, st));
        us_enqueue += since(t_mark); t_mark = std::chrono::steady_clock::now();
        HIP_TRY(c, hipStreamSynchronize(st));
        us_wait += since(t_mark); t_mark = std::chrono::steady_clock::now();
        all_done = true;
        for (int b = 0; b < B; ++b) {
            if (was_done[b]) continue;
            const int* cb = ctl_h + (size_t)b * 4;
            const int ran = cb[0] ? std::min(nb, cb[3] - s0 + 1) : nb;
            if (ran > 0) elbo[b] = hist_h[(size_t)b * (SB_K + 4) + ran - 1];
            iters[b] = cb[1];
            conv[b] = cb[2];
            for (int k = 0; k < 3 * G && !info[b]; ++k)
                if (info_h[(size_t)b * 3 * G + k] > 0) info[b] = info_h[(size_t)b * 3 * G + k];
            if (info[b]) elbo[b] = NAN;                       // (jax's cholesky: NaN from the failed pivot on, no exception)
            if (cb[0]) was_done[b] = 1;
            else all_done = false;
        }
        us_host += since(t_mark); t_mark = std::chrono::steady_clock::now();
    }
    if (mu_out && var_out) {
        for (int k = 0; k < 4; ++k)                            // (B evaluations of each copy, not the buffers' capacity)
            HIP_TRY(c, hipMemcpyAsync(st_h + (size_t)k * cap * d, m->state + (size_t)k * cap * d, (size_t)B * d * sizeof(double),
                                      hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) {
            const bool in_b = iters[b] >= 1 && (iters[b] & 1);     // odd trips wrote copy B
            memcpy(mu_out + (size_t)b * d, st_h + ((in_b ? 2 : 0) * (size_t)cap + b) * d, d * sizeof(double));
            memcpy(var_out + (size_t)b * d, st_h + ((in_b ? 3 : 1) * (size_t)cap + b) * d, d * sizeof(double));
        }
    }
    if (timers_env)
        fprintf(stderr, "[gprn] elbocalc_batch (one tile), %d evaluations, us: buffers %.0f | staging %.0f | enqueue %.0f | waiting for the device "
                        "%.0f | verdicts %.0f | states back %.0f | total %.0f (%d launches of sweeps)\n", B, us_ensure, us_stage, us_enqueue,
                us_wait, us_host, since(t_mark), since(t_begin), s - (max_iter >= 1 ? 1 : 0));
    return GPRN_OK;
}
