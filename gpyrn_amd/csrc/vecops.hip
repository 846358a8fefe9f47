// O(N) and O(N^2) kernels around the factorisation: everything of one ELBOaux
// (meanfield.py:651-710) that is not an N^3 contraction.  All batched over the
// latent GPs of a phase ("slots") through grid.y / grid.z; HBM-bound.
//
//   k_prep_nodes / k_prep_weights   d, sqrt(d), right-hand side, q = rhs/sqrt(d)   meanfield.py:759-791, 838-864
//   k_build_B                       B = I + D^1/2 K D^1/2 (lower tiles)
//   k_logdet                        2 sum log diag(L)             meanfield.py:1029,1062,1088,1091
//   k_lower_matvec                  y = M v, M lower triangular
//   k_colops_partial / _reduce      colnorm^2(X) = diag(B^-1),  X^T u
//   k_finalize                      new mu, new var = diag Sigma, tr(B^-1)
//   k_q1_rows / k_sum_rows          <K_j^-1, Sigma_k> for the cumulative-trace quirk, :1039-1041
//   k_dot_self                      a.a  (mu^T K^-1 mu via a = L_K^-1 mu, :1032,1050)
//   k_loglike_partial / k_elbo_final  expected log-likelihood (:895-990) + assembly (:709)
#include "gprn_internal.h"
#include "vecops.h"

#include <math.h>

#include <algorithm>

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over a 256-thread block; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 4 doubles */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return r;
}

__global__ __launch_bounds__(256)
void k_prep_nodes(const int* __restrict__ slot_gp, int N, int ld, int p, int q,
                  const double* __restrict__ mu, const double* __restrict__ var,
                  const double* __restrict__ yres, const double* __restrict__ variance,
                  double* __restrict__ d, double* __restrict__ s, double* __restrict__ pred,
                  double* __restrict__ z, EvalMap ev)
{
    const int slot = blockIdx.y, j = slot_gp[slot];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= ld) return;
    const size_t eb = ev_of(ev, slot);               // (several evaluations side by side: this slot's copy of the problem)
    mu += eb * ev.state; var += eb * ev.state; yres += eb * ev.yv; variance += eb * ev.yv;
    double dsum = 1.0, psum = 0.0;
    if (n < N) {
        dsum = 0.0;
        for (int i = 0; i < p; ++i) {
            const double vi = variance[(size_t)i * N + n];
            const size_t wrow = (size_t)(1 + i) * q;
            const double mwj = mu[(wrow + j) * N + n];
            const double vwj = var[(wrow + j) * N + n];
            dsum += (mwj * mwj + vwj) / vi;
            double other = 0.0;
            for (int k = 0; k < q; ++k)
                if (k != j) other += mu[(wrow + k) * N + n] * mu[(size_t)k * N + n];
            psum += (yres[(size_t)i * N + n] - other) * mwj / vi;
        }
    }
    const size_t o = (size_t)slot * ld + n;
    d[o] = dsum;
    s[o] = sqrt(dsum);
    pred[o] = psum;
    z[o] = psum / sqrt(dsum);           // q = D^-1/2 pred: Sigma pred = D^-1/2 (q - X^T X q)
}

__global__ __launch_bounds__(256)
void k_prep_weights(const int* __restrict__ slot_gp, int N, int ld, int p, int q,
                    const double* __restrict__ mu, const double* __restrict__ var,
                    const double* __restrict__ yres, const double* __restrict__ variance,
                    double* __restrict__ d, double* __restrict__ s, double* __restrict__ pred,
                    double* __restrict__ z, EvalMap ev)
{
    const int slot = blockIdx.y, kk = slot_gp[slot] - q;
    const int j = kk / p, i = kk % p;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= ld) return;
    const size_t eb = ev_of(ev, slot);
    mu += eb * ev.state; var += eb * ev.state; yres += eb * ev.yv; variance += eb * ev.yv;
    double dv = 1.0, pv = 0.0;
    if (n < N) {
        const double vi = variance[(size_t)i * N + n];
        const double mfj = mu[(size_t)j * N + n];
        dv = (mfj * mfj + var[(size_t)j * N + n]) / vi;
        const size_t wrow = (size_t)(1 + i) * q;
        double other = 0.0;
        for (int k = 0; k < q; ++k)
            if (k != j) other += mu[(size_t)k * N + n] * mu[(wrow + k) * N + n];
        pv = (yres[(size_t)i * N + n] - other) * mfj / vi;
    }
    const size_t o = (size_t)slot * ld + n;
    d[o] = dv;
    s[o] = sqrt(dv);
    pred[o] = pv;
    z[o] = pv / sqrt(dv);
}

// B = I + D^1/2 K D^1/2 on the lower tiles (diagonal tiles in full); identity padding
// part 0: every lower tile; part 1: what the factorisation's first outer panel touches before its K = outer * 128
// update (tile columns < outer, and of the next `outer` columns the diagonal and sub-diagonal tiles, which the in-panel
// lists keep up to date step by step: ensure_tasks, factor.hip); part 2: the others (built beside the first tile steps)
// One workgroup per tile that is written, no empty ones: the grid's x index runs over the tiles of the part, column by
// column (a workgroup that finds out it has nothing to do still costs its dispatch -- with T x T workgroups of which a
// quarter of the lower half had work, part 1 took 139 us for 45 us of traffic at config 3).
__device__ __forceinline__ bool build_B_tile(int idx, int T, int part, int outer, int& ti, int& tj)
{
    if (part == 1) {
        // columns [0, outer): rows tj .. T-1;  then columns [outer, 2 outer): rows tj, tj + 1
        for (tj = 0; tj < outer && tj < T; ++tj) {
            const int n = T - tj;
            if (idx < n) { ti = tj + idx; return true; }
            idx -= n;
        }
        for (; tj < 2 * outer && tj < T; ++tj) {
            const int n = tj + 1 < T ? 2 : 1;
            if (idx < n) { ti = tj + idx; return true; }
            idx -= n;
        }
        return false;
    }
    // part 0: every lower tile; part 2: the lower tiles part 1 leaves out -- row by row
    for (ti = 0; ti < T; ++ti) {
        if (idx <= ti) {
            tj = idx;
            if (part == 2 && (tj < outer || (tj < 2 * outer && ti <= tj + 1))) return false;
            return true;
        }
        idx -= ti + 1;
    }
    return false;
}

__global__ __launch_bounds__(256)
void k_build_B(double* const* __restrict__ ptrs, int N, int ld, const double* __restrict__ s, int part, int outer, int T)
{
    int ti, tj;
    const int slot = blockIdx.y;
    if (!build_B_tile((int)blockIdx.x, T, part, outer, ti, tj)) return;
    const double* K = ptrs[(size_t)slot * GPRN_NBUF + BUF_K];
    double* B = ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    const double* sv = s + (size_t)slot * ld;
    const int n = tj * GPRN_TILE + 2 * (threadIdx.x & 63);
    const double s0 = sv[n], s1 = sv[n + 1];
    // four rows per pass, all loads issued before the first store: a thread keeps 64 B in flight
    // instead of 16 (the kernel is latency-limited otherwise: 3.4 TB/s)
    for (int it0 = 0; it0 < 32; it0 += 4) {
        v2d kv[4];
        double sm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = ti * GPRN_TILE + (threadIdx.x >> 6) + 4 * (it0 + u);
            const bool in = m < N;
            kv[u] = in ? *reinterpret_cast<const v2d*>(K + (size_t)m * ld + n) : v2d{0.0, 0.0};
            sm[u] = in ? sv[m] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = ti * GPRN_TILE + (threadIdx.x >> 6) + 4 * (it0 + u);
            v2d out;
            out.x = (n < N) ? sm[u] * s0 * kv[u].x : 0.0;
            out.y = (n + 1 < N) ? sm[u] * s1 * kv[u].y : 0.0;
            if (m == n) out.x += 1.0;
            if (m == n + 1) out.y += 1.0;
            *reinterpret_cast<v2d*>(B + (size_t)m * ld + n) = out;
        }
    }
}

// out[idx(slot)] = 2 * sum_{n<N} log M[n][n]
__global__ __launch_bounds__(256)
void k_logdet(double* const* __restrict__ ptrs, int buf, int N, int ld,
              const int* __restrict__ slot_gp, double* __restrict__ out, EvalMap ev, size_t out_stride)
{
    __shared__ double sh[4];
    const int slot = blockIdx.x;
    const double* M = ptrs[(size_t)slot * GPRN_NBUF + buf];
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) acc += log(M[(size_t)n * ld + n]);
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[ev_of(ev, slot) * out_stride + slot_gp[slot]] = 2.0 * acc;
}

// out[slot][i] = sum_{c<=i} M[i][c] v[c]   (i < N), M = ptrs[slot][buf];
// v = vin + (vin_by_gp ? slot_gp[slot] : slot) * vstride; rows row0 + 4 * blockIdx.x .. of the matrix
__global__ __launch_bounds__(256)
void k_lower_matvec(double* const* __restrict__ ptrs, int buf, int N, int ld,
                    const double* __restrict__ vin, size_t vstride, int vin_by_gp,
                    const int* __restrict__ slot_gp, double* __restrict__ out, int row0, EvalMap ev)
{
    const int slot = blockIdx.y;
    const int i = row0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= ld) return;
    const double* M = ptrs[(size_t)slot * GPRN_NBUF + buf] + (size_t)i * ld;
    // (vin_by_gp: a row of the state -- of this slot's evaluation)
    const double* v = vin + (vin_by_gp ? ev_of(ev, slot) * ev.state + (size_t)slot_gp[slot] * vstride : (size_t)slot * vstride);
    double acc = 0.0;
    if (i < N) {
        for (int c = 2 * lane; c <= i; c += 128) {
            const v2d mv = *reinterpret_cast<const v2d*>(M + c);
            acc += mv.x * v[c];
            if (c + 1 <= i) acc += mv.y * v[c + 1];
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)slot * ld + i] = acc;
}

// partial column sums over one tile row (128 rows) of X: cs += x^2, ct += x*u[row]
// grid (ld/64, T, nslots); tiles above the diagonal are skipped (and not read later)
__global__ __launch_bounds__(256)
void k_colops_partial(double* const* __restrict__ ptrs, int ld, int T,
                      const double* __restrict__ u, double* __restrict__ part, int ch0)
{
    __shared__ double shs[4][64], sht[4][64];
    const int c0 = blockIdx.x * 64, ch = ch0 + blockIdx.y, slot = blockIdx.z;
    if (ch < (c0 >> 7)) return;
    const double* X = ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    const double* uv = u + (size_t)slot * ld;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double cs = 0.0, ct = 0.0;
    // eight rows per pass, loads first (latency-limited otherwise); accumulation order unchanged
    for (int r0 = ch * GPRN_TILE + rl; r0 < (ch + 1) * GPRN_TILE; r0 += 32) {
        double x[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[k] = X[(size_t)(r0 + 4 * k) * ld + c0 + cl];
            w[k] = uv[r0 + 4 * k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cs += x[k] * x[k];
            ct += x[k] * w[k];
        }
    }
    shs[rl][cl] = cs;
    sht[rl][cl] = ct;
    __syncthreads();
    if (rl == 0) {
        const size_t o = (((size_t)slot * T + ch) * 2) * ld + c0 + cl;
        part[o] = (shs[0][cl] + shs[1][cl]) + (shs[2][cl] + shs[3][cl]);
        part[o + ld] = (sht[0][cl] + sht[1][cl]) + (sht[2][cl] + sht[3][cl]);
    }
}

__global__ __launch_bounds__(256)
void k_colops_reduce(int ld, int T, const double* __restrict__ part,
                     double* __restrict__ cs, double* __restrict__ ct)
{
    const int slot = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ld) return;
    double a = 0.0, b = 0.0;
    for (int ch = c >> 7; ch < T; ++ch) {
        const size_t o = (((size_t)slot * T + ch) * 2) * ld + c;
        a += part[o];
        b += part[o + ld];
    }
    cs[(size_t)slot * ld + c] = a;
    ct[(size_t)slot * ld + c] = b;
}

// new mu = Sigma pred = (q - X^T X q)/s  (q = pred/s: no product with K needed),
// new var = (1 - diag B^-1)/d into the state rows of the GP;
// trBinv[gp] = sum diag B^-1
__global__ __launch_bounds__(256)
void k_finalize(const int* __restrict__ slot_gp, int N, int ld, int p, int q,
                const double* __restrict__ d, const double* __restrict__ s,
                const double* __restrict__ z, const double* __restrict__ cs, const double* __restrict__ ct,
                double* __restrict__ mu, double* __restrict__ var, double* __restrict__ trBinv,
                double* const* __restrict__ ptrs, double* __restrict__ logdetB, EvalMap ev)
{
    __shared__ double sh[4];
    const int slot = blockIdx.x, gp = slot_gp[slot];
    const size_t eb = ev_of(ev, slot);
    mu += eb * ev.state; var += eb * ev.state; trBinv += eb * ev.scal; logdetB += eb * ev.scal;
    size_t row;
    if (gp < q) row = gp;
    else { const int kk = gp - q, j = kk / p, i = kk % p; row = (size_t)(1 + i) * q + j; }
    // log det B = 2 sum log diag(L) in the same pass (k_logdet's sum, same order: one launch less on the phase's tail)
    const double* Lm = ptrs ? ptrs[(size_t)slot * GPRN_NBUF + BUF_B] : nullptr;
    double tr = 0.0, ld_acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const size_t o = (size_t)slot * ld + n;
        const double binv = cs[o];
        tr += binv;
        mu[row * N + n] = (z[o] - ct[o]) / s[o];
        var[row * N + n] = (1.0 - binv) / d[o];
        if (Lm) ld_acc += log(Lm[(size_t)n * ld + n]);
    }
    tr = block_sum(tr, sh);
    if (threadIdx.x == 0) trBinv[gp] = tr;
    if (Lm) {
        ld_acc = block_sum(ld_acc, sh);
        if (threadIdx.x == 0) logdetB[gp] = 2.0 * ld_acc;
    }
}

// k_colops_reduce and k_finalize in one launch of ld / 256 workgroups per GP (16 at N = 4096) instead of two launches, the
// second of ONE workgroup per GP whose threads walk 16 elements each through a dependent chain of loads, a strided
// diagonal read, a log and two divisions (21-25 us at the end of every phase, on the chain stream, with nothing beside
// it).  A thread owns one element: column sums in k_colops_reduce's order, mu / var, and the element's two terms of
// tr B^-1 and log det B go to `terms`; the LAST workgroup of a GP to finish (ticket counter, reset by it) adds them up
// exactly as k_finalize's threads did -- thread t: n = t, t + 256, ... in turn, then block_sum -- so both scalars keep
// their bits.
// The hand-over of the terms to that last workgroup, FENCED = false (the default on gfx942 / gfx950, GPRN_RELAXED_HANDOVER):
// relaxed agent-scope stores, s_waitcnt vmcnt(0), a relaxed ticket, relaxed agent-scope loads -- no release / acquire pair,
// so by the letter of the HIP memory model nothing orders the terms before the ticket.  What does, on this hardware:
//   1. an agent-scope atomic store is issued with sc1: it is WRITTEN THROUGH this XCD's L2 to memory (the L2s of the eight
//      XCDs are not coherent with each other; agent scope is the scope that crosses them), and the store's acknowledgement
//      -- which s_waitcnt vmcnt(0) waits for: stores count in vmcnt on gfx9 -- comes back only after that;
//   2. the ticket's read-modify-write is issued after the barrier behind that wait (a wave issues its memory instructions
//      in order, and the asm statement is a compiler barrier), and is performed AT memory (device-scope atomics execute in
//      the memory-side L2 / fabric, not in an XCD-local line);
//   3. the last workgroup's loads are issued after the ticket's value has returned (the branch depends on it) and, being
//      agent-scope atomic loads (sc1), bypass its own XCD's L2 lines: they read what 1. wrote.
// The release / acquire form (FENCED; what any other --offload-arch gets) costs an L2 write-back per workgroup: 84 us for
// the 192 matrices of a batch's weight phase against 9.  tests/test_parity_gpu.py::
// test_reduce_finalize_relaxed_handover_keeps_the_bits runs both forms 200 times over more workgroups per slot than an
// XCD holds and compares tr B^-1 and log det B bit for bit.
template <bool FENCED>
__global__ __launch_bounds__(256)
void k_reduce_finalize(const int* __restrict__ slot_gp, int N, int ld, int T, int p, int q,
                       const double* __restrict__ part, const double* __restrict__ d, const double* __restrict__ s,
                       const double* __restrict__ z, double* __restrict__ cs, double* __restrict__ ct,
                       double* __restrict__ mu, double* __restrict__ var, double* __restrict__ trBinv,
                       double* const* __restrict__ ptrs, double* __restrict__ logdetB,
                       double* __restrict__ terms /* [slot][2][ld] */, unsigned* __restrict__ tickets /* [slot] */, EvalMap ev)
{
    __shared__ double sh[4];
    __shared__ unsigned last;
    const int slot = blockIdx.y, gp = slot_gp[slot], n = blockIdx.x * 256 + threadIdx.x;
    const size_t eb = ev_of(ev, slot);
    mu += eb * ev.state; var += eb * ev.state; trBinv += eb * ev.scal; logdetB += eb * ev.scal;
    size_t row;
    if (gp < q) row = gp;
    else { const int kk = gp - q, j = kk / p, i = kk % p; row = (size_t)(1 + i) * q + j; }
    const double* Lm = ptrs ? ptrs[(size_t)slot * GPRN_NBUF + BUF_B] : nullptr;
    double* const tt = terms + (size_t)slot * 2 * ld;
    if (n < ld) {
        double a = 0.0, b = 0.0;
        for (int ch = n >> 7; ch < T; ++ch) {
            const size_t o = (((size_t)slot * T + ch) * 2) * ld + n;
            a += part[o];
            b += part[o + ld];
        }
        const size_t o = (size_t)slot * ld + n;
        cs[o] = a;
        ct[o] = b;
        if (n < N) {
            mu[row * N + n] = (z[o] - b) / s[o];
            var[row * N + n] = (1.0 - a) / d[o];
            // (the terms go to the last workgroup of this launch, possibly on another XCD: agent-scope stores -- written
            // through -- and loads, no fence.  A __threadfence() here is an L2 write-back per workgroup, on an L2 full of
            // the factorisation's freshly written tiles: 84 us for the 192 matrices of a batch's weight phase)
            __hip_atomic_store(tt + n, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (Lm) __hip_atomic_store(tt + ld + n, log(Lm[(size_t)n * ld + n]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    constexpr bool fenced = FENCED || !GPRN_RELAXED_HANDOVER;
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every thread's stores acknowledged, then the workgroup's ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned got = fenced ? __hip_atomic_fetch_add(tickets + slot, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                    : __hip_atomic_fetch_add(tickets + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = got + 1 == gridDim.x ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;                              // (uniform)
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double tr = 0.0, ld_acc = 0.0;
    for (int m = threadIdx.x; m < N; m += 256) {
        tr += __hip_atomic_load(tt + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (Lm) ld_acc += __hip_atomic_load(tt + ld + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tr = block_sum(tr, sh);
    if (threadIdx.x == 0) { trBinv[gp] = tr; atomicExch(tickets + slot, 0u); }
    if (Lm) {
        ld_acc = block_sum(ld_acc, sh);
        if (threadIdx.x == 0) logdetB[gp] = 2.0 * ld_acc;
    }
}

// rowsum[m] = sum_{n<=m} w(m,n) Kinv[m][n] (delta_mn - Binv[m][n]) / (s_m s_n), w = 2 off-diagonal
__global__ __launch_bounds__(256)
void k_q1_rows(const double* __restrict__ Kinv, const double* __restrict__ Binv, int N, int ld,
               const double* __restrict__ s, double* __restrict__ rowsum)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= N) return;
    const double* kr = Kinv + (size_t)m * ld;
    const double* br = Binv + (size_t)m * ld;
    double acc = 0.0;
    for (int n = lane; n < m; n += 64) acc -= kr[n] * br[n] / s[n];
    acc = wave_sum(acc);
    if (lane == 0) {
        const double sm = s[m];
        rowsum[m] = 2.0 * acc / sm + kr[m] * (1.0 - br[m]) / (sm * sm);
    }
}

__global__ __launch_bounds__(256)
void k_sum_to(const double* __restrict__ v, int n, double* __restrict__ out)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) *out = acc;
}

__global__ __launch_bounds__(256)
void k_dot_self(const int* __restrict__ slot_gp, int N, int ld, const double* __restrict__ a,
                double* __restrict__ out, EvalMap ev)
{
    __shared__ double sh[4];
    const int slot = blockIdx.x;
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const double x = a[(size_t)slot * ld + n];
        acc += x * x;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[ev_of(ev, slot) * ev.scal + slot_gp[slot]] = acc;
}

// Expected log-likelihood (meanfield.py:895-990; y_raw is the RAW data, quirk Q3): partial sums
// of its three terms over a slice of the time stamps per block (fixed slices -> deterministic).
#define ELBO_BLOCKS 32
__global__ __launch_bounds__(256)
void k_loglike_partial(int N, int p, int q, const double* __restrict__ mu, const double* __restrict__ var,
                       const double* __restrict__ yraw, const double* __restrict__ variance,
                       double* __restrict__ part /* [ELBO_BLOCKS][3] */, const int* __restrict__ evals, EvalMap ev)
{
    __shared__ double sh[4];
    const double TWO_PI = 6.283185307179586;
    // (several evaluations side by side: grid y = position in the list `evals` of evaluations still running)
    const size_t eb = evals ? (size_t)evals[blockIdx.y] : 0;
    mu += eb * ev.state; var += eb * ev.state; variance += eb * ev.yv;
    part += (size_t)blockIdx.y * 3 * ELBO_BLOCKS;
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += 256 * ELBO_BLOCKS) {
        for (int i = 0; i < p; ++i) {
            const double vi = variance[(size_t)i * N + n];
            const size_t wrow = (size_t)(1 + i) * q;
            t1 += log(TWO_PI * vi);
            double fit = 0.0, cross = 0.0;
            for (int j = 0; j < q; ++j) {
                const double mf = mu[(size_t)j * N + n], vf = var[(size_t)j * N + n];
                const double mw = mu[(wrow + j) * N + n], vw = var[(wrow + j) * N + n];
                fit += mw * mf;
                cross += vf * (mw * mw) + vw * (mf * mf) + vf * vw;
            }
            const double resid = yraw[(size_t)i * N + n] - fit;
            t2 += resid * resid / vi;
            t3 += cross / vi;
        }
    }
    t1 = block_sum(t1, sh);
    t2 = block_sum(t2, sh);
    t3 = block_sum(t3, sh);
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = t1;
        part[3 * blockIdx.x + 1] = t2;
        part[3 * blockIdx.x + 2] = t3;
    }
}

// Final assembly  ELBO = (LogL + LogP + Ent) / q  (meanfield.py:709, :1023-1065, :1085-1093).
// out[0..3] = ELBO, LogL, LogP, Ent.
__global__ void k_elbo_final(int N, int p, int q, const double* __restrict__ part,
                             const double* __restrict__ logdetK, const double* __restrict__ logdetB,
                             const double* __restrict__ trBinv, const double* __restrict__ muKmu,
                             const double* __restrict__ q1, double* __restrict__ out, const int* __restrict__ evals, EvalMap ev)
{
    if (threadIdx.x != 0) return;
    const double TWO_PI = 6.283185307179586;
    const size_t eb = evals ? (size_t)evals[blockIdx.x] : 0;
    part += (size_t)blockIdx.x * 3 * ELBO_BLOCKS;
    logdetK += eb * ev.G; logdetB += eb * ev.scal; trBinv += eb * ev.scal; muKmu += eb * ev.scal; q1 += eb * ev.scal;
    out += eb * 4;
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    for (int b = 0; b < ELBO_BLOCKS; ++b) { t1 += part[3 * b]; t2 += part[3 * b + 1]; t3 += part[3 * b + 2]; }
    const int G = q + q * p;
    const double logl = -0.5 * t1 - 0.5 * t2 - 0.5 * t3;
    double ent = 0.0, logp = 0.0;
    for (int g = 0; g < G; ++g) {
        ent += 0.5 * (logdetK[g] - logdetB[g]);
        double tr = trBinv[g];
        if (g < q)
            for (int k = 0; k < g; ++k) tr += q1[g * q + k];   // cumulative sumSigmaF, quirk Q1
        logp += -0.5 * logdetK[g] - 0.5 * (muKmu[g] + tr);
    }
    const double c = (double)q * (p + 1) * N;
    ent += 0.5 * c * (1.0 + log(TWO_PI));
    logp += -0.5 * c * log(TWO_PI);
    out[0] = (logl + logp + ent) / q;
    out[1] = logl;
    out[2] = logp;
    out[3] = ent;
}

// ------------------------------------------------------------------ launchers
#define LAUNCH_END(c) do { prof_end(c); HIP_TRY(c, hipGetLastError()); return GPRN_OK; } while (0)

int vec_prep(gprn_ctx* c, bool weights, const int* d_slot_gp, int nslots)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    const size_t o = (size_t)c->slot0 * c->ld;
    dim3 grid((c->ld + 255) / 256, nslots);
    if (weights)
        hipLaunchKernelGGL(k_prep_weights, grid, dim3(256), 0, c->stream, d_slot_gp, c->N, c->ld,
                           c->p, c->q, c->d_mu, c->d_var, c->d_yres, c->d_variance,
                           c->d_d + o, c->d_s + o, c->d_pred + o, c->d_z + o, c->ev);
    else
        hipLaunchKernelGGL(k_prep_nodes, grid, dim3(256), 0, c->stream, d_slot_gp, c->N, c->ld,
                           c->p, c->q, c->d_mu, c->d_var, c->d_yres, c->d_variance,
                           c->d_d + o, c->d_s + o, c->d_pred + o, c->d_z + o, c->ev);
    LAUNCH_END(c);
}

int vec_build_B(gprn_ctx* c, int nslots, hipStream_t stream, int part, int outer)
{
    if (!nslots) return GPRN_OK;
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_BUILD_B, stream);
    const int T = c->T;
    int ntiles = T * (T + 1) / 2;                              // parts 0 and 2 (part 2's few part-1 tiles exit at once)
    if (part == 1) {
        ntiles = 0;
        for (int tj = 0; tj < outer && tj < T; ++tj) ntiles += T - tj;
        for (int tj = outer; tj < 2 * outer && tj < T; ++tj) ntiles += tj + 1 < T ? 2 : 1;
    }
    hipLaunchKernelGGL(k_build_B, dim3(ntiles, nslots), dim3(256), 0, stream,
                       (double* const*)c->d_ptrs, c->N, c->ld, c->d_s + (size_t)c->slot0 * c->ld, part, outer, T);
    LAUNCH_END(c);
}

int vec_logdet(gprn_ctx* c, int buf, const int* d_slot_gp, int nslots, double* out)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_logdet, dim3(nslots), dim3(256), 0, c->stream,
                       (double* const*)c->d_ptrs, buf, c->N, c->ld, d_slot_gp, out, c->ev,
                       out == c->d_logdetK ? c->ev.G : c->ev.scal);
    LAUNCH_END(c);
}

int vec_lower_matvec(gprn_ctx* c, int buf, const double* vin, size_t vstride, int vin_by_gp,
                     const int* d_slot_gp, int nslots, double* out, hipStream_t stream, int row0, int nrows)
{
    if (!nslots) return GPRN_OK;
    if (!stream) stream = c->stream;
    if (nrows < 0) nrows = c->ld - row0;
    if (nrows <= 0) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_lower_matvec, dim3((nrows + 3) / 4, nslots), dim3(256), 0, stream,
                       (double* const*)c->d_ptrs, buf, c->N, c->ld, vin, vstride, vin_by_gp,
                       d_slot_gp, out, row0, c->ev);
    LAUNCH_END(c);
}

// partial column sums of tile rows [ch0, ch0 + nch) (nch < 0: to the last); columns right of tile row ch0 + nch - 1
// hold nothing of these rows
int vec_colops_partial(gprn_ctx* c, int nslots, hipStream_t stream, int ch0, int nch)
{
    if (!nslots) return GPRN_OK;
    if (!stream) stream = c->stream;
    if (nch < 0) nch = c->T - ch0;
    if (nch <= 0) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC, stream);
    const size_t o = (size_t)c->slot0 * c->ld, po = (size_t)c->slot0 * c->T * 2 * c->ld;
    const int ncol64 = std::min(c->ld / 64, 2 * (ch0 + nch));
    hipLaunchKernelGGL(k_colops_partial, dim3(ncol64, nch, nslots), dim3(256), 0, stream,
                       (double* const*)c->d_ptrs, c->ld, c->T, c->d_u + o, c->d_part + po, ch0);
    LAUNCH_END(c);
}

int vec_colops_reduce(gprn_ctx* c, int nslots)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    const size_t o = (size_t)c->slot0 * c->ld, po = (size_t)c->slot0 * c->T * 2 * c->ld;
    hipLaunchKernelGGL(k_colops_reduce, dim3((c->ld + 255) / 256, nslots), dim3(256), 0,
                       c->stream, c->ld, c->T, c->d_part + po, c->d_cs + o, c->d_ct + o);
    LAUNCH_END(c);
}

int vec_colops(gprn_ctx* c, int nslots)
{
    int rc = vec_colops_partial(c, nslots, nullptr, 0, -1);
    return rc ? rc : vec_colops_reduce(c, nslots);
}

int vec_finalize(gprn_ctx* c, const int* d_slot_gp, int nslots, bool with_logdet)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    const size_t o = (size_t)c->slot0 * c->ld;
    hipLaunchKernelGGL(k_finalize, dim3(nslots), dim3(256), 0, c->stream, d_slot_gp, c->N, c->ld,
                       c->p, c->q, c->d_d + o, c->d_s + o, c->d_z + o, c->d_cs + o, c->d_ct + o, c->d_mu, c->d_var,
                       c->d_trBinv, with_logdet ? (double* const*)c->d_ptrs : (double* const*)nullptr, c->d_logdetB, c->ev);
    LAUNCH_END(c);
}

// vec_colops_reduce + vec_finalize (with log det B) in one launch
int vec_reduce_finalize(gprn_ctx* c, const int* d_slot_gp, int nslots, bool with_logdet)
{
    if (!nslots) return GPRN_OK;
    if (!c->d_fin_terms) {
        HIP_TRY(c, hipMalloc(&c->d_fin_terms, (size_t)c->nslot * 2 * c->ld * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_fin_tickets, (size_t)c->nslot * sizeof(unsigned)));
        HIP_TRY(c, hipMemset(c->d_fin_tickets, 0, (size_t)c->nslot * sizeof(unsigned)));
    }
    prof_begin(c, GPRN_T_VEC);
    const size_t o = (size_t)c->slot0 * c->ld, po = (size_t)c->slot0 * c->T * 2 * c->ld;
#define GO_RF(F) hipLaunchKernelGGL(k_reduce_finalize<F>, dim3((c->ld + 255) / 256, nslots), dim3(256), 0, c->stream, d_slot_gp, c->N, c->ld, c->T, \
                       c->p, c->q, c->d_part + po, c->d_d + o, c->d_s + o, c->d_z + o, c->d_cs + o, c->d_ct + o, c->d_mu, c->d_var, \
                       c->d_trBinv, with_logdet ? (double* const*)c->d_ptrs : (double* const*)nullptr, c->d_logdetB, \
                       c->d_fin_terms + (size_t)c->slot0 * 2 * c->ld, c->d_fin_tickets + c->slot0, c->ev)
    if (c->fenced_finalize) GO_RF(true); else GO_RF(false);
#undef GO_RF
    LAUNCH_END(c);
}

int vec_q1(gprn_ctx* c, const double* Kinv_j, const double* Binv_k, const double* s_k,
           double* scratch, double* out_scalar, hipStream_t stream)
{
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_q1_rows, dim3((c->N + 3) / 4), dim3(256), 0, stream, Kinv_j, Binv_k,
                       c->N, c->ld, s_k, scratch);
    hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, stream, scratch, c->N, out_scalar);
    LAUNCH_END(c);
}

int vec_dot_self(gprn_ctx* c, const int* d_slot_gp, int nslots, const double* a, double* out, hipStream_t stream)
{
    if (!nslots) return GPRN_OK;
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_dot_self, dim3(nslots), dim3(256), 0, stream, d_slot_gp, c->N, c->ld,
                       a, out, c->ev);
    LAUNCH_END(c);
}

// scal: the sweep's per-GP scalars (logdetB | trBinv | muKmu | q1); part: 3 * ELBO_BLOCKS doubles of scratch
int vec_elbo(gprn_ctx* c, double* out4, const double* scal, double* part, hipStream_t stream)
{
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_loglike_partial, dim3(ELBO_BLOCKS), dim3(256), 0, stream, c->N, c->p, c->q,
                       c->d_mu, c->d_var, c->d_yraw, c->d_variance, part, (const int*)nullptr, c->ev);
    hipLaunchKernelGGL(k_elbo_final, dim3(1), dim3(64), 0, stream, c->N, c->p, c->q, part,
                       c->d_logdetK, scal, scal + c->G, scal + 2 * (size_t)c->G, scal + 3 * (size_t)c->G, out4,
                       (const int*)nullptr, c->ev);
    LAUNCH_END(c);
}

// the same for the n evaluations listed in d_evals (midn.hip): evaluation b reads its own state, variance and per-GP scalars
// (strides c->ev) and writes out4 + 4 b; part: n x 3 * ELBO_BLOCKS doubles of scratch
int vec_elbo_evals(gprn_ctx* c, const int* d_evals, int n, double* out4, const double* scal, double* part, hipStream_t stream)
{
    if (!n) return GPRN_OK;
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_loglike_partial, dim3(ELBO_BLOCKS, n), dim3(256), 0, stream, c->N, c->p, c->q,
                       c->d_mu, c->d_var, c->d_yraw, c->d_variance, part, d_evals, c->ev);
    hipLaunchKernelGGL(k_elbo_final, dim3(n), dim3(64), 0, stream, c->N, c->p, c->q, part,
                       c->d_logdetK, scal, scal + c->G, scal + 2 * (size_t)c->G, scal + 3 * (size_t)c->G, out4, d_evals, c->ev);
    LAUNCH_END(c);
}

// quirk Q1 for the evaluations of a batch (midn.hip; node slots node-major: slot = k * n_eval + a): per (node k, node j > k,
// position a) the row sums of k_q1_rows and their sum (k_sum_to's order) into that evaluation's q1[j * q + k]
__global__ __launch_bounds__(256)
void k_q1_rows_evals(double* const* __restrict__ ptrs, const int* __restrict__ slot_eval, const double* __restrict__ Kinv_slab,
                     size_t nn, int q, int n_eval, int N, int ld, const double* __restrict__ s, double* __restrict__ rowsum)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= N) return;
    int kj = blockIdx.y / n_eval, k = 0, j = 1;
    const int a = blockIdx.y % n_eval;
    while (kj >= q - 1 - k) { kj -= q - 1 - k; ++k; }
    j = k + 1 + kj;
    const int slot = k * n_eval + a;
    const size_t b = (size_t)slot_eval[slot];
    const double* kr = Kinv_slab + (b * (size_t)(q - 1) + (size_t)(j - 1)) * nn + (size_t)m * ld;
    const double* br = ptrs[(size_t)slot * GPRN_NBUF + BUF_B] + (size_t)m * ld;
    const double* sv = s + (size_t)slot * ld;
    double acc = 0.0;
    for (int n = lane; n < m; n += 64) acc -= kr[n] * br[n] / sv[n];
    acc = wave_sum(acc);
    if (lane == 0) {
        const double sm = sv[m];
        rowsum[(size_t)blockIdx.y * ld + m] = 2.0 * acc / sm + kr[m] * (1.0 - br[m]) / (sm * sm);
    }
}

__global__ __launch_bounds__(256)
void k_q1_sum_evals(const double* __restrict__ rowsum, const int* __restrict__ slot_eval, int q, int n_eval, int N, int ld,
                    double* __restrict__ q1, size_t scal_stride)
{
    __shared__ double sh[4];
    int kj = blockIdx.x / n_eval, k = 0;
    const int a = blockIdx.x % n_eval;
    while (kj >= q - 1 - k) { kj -= q - 1 - k; ++k; }
    const int j = k + 1 + kj;
    const double* v = rowsum + (size_t)blockIdx.x * ld;
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += v[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) q1[(size_t)slot_eval[k * n_eval + a] * scal_stride + (size_t)j * q + k] = acc;
}

int vec_q1_evals(gprn_ctx* c, const int* d_slot_eval, const double* Kinv_slab, int n_eval, double* scratch, hipStream_t stream)
{
    const int npair = c->q * (c->q - 1) / 2;
    if (!n_eval || !npair) return GPRN_OK;
    if (!stream) stream = c->stream;
    prof_begin(c, GPRN_T_VEC, stream);
    hipLaunchKernelGGL(k_q1_rows_evals, dim3((c->N + 3) / 4, npair * n_eval), dim3(256), 0, stream, (double* const*)c->d_ptrs,
                       d_slot_eval, Kinv_slab, (size_t)c->ld * c->ld, c->q, n_eval, c->N, c->ld, c->d_s, scratch);
    hipLaunchKernelGGL(k_q1_sum_evals, dim3(npair * n_eval), dim3(256), 0, stream, (const double*)scratch, d_slot_eval, c->q,
                       n_eval, c->N, c->ld, c->d_q1, c->ev.scal);
    LAUNCH_END(c);
}

// Sigma = D^-1/2 (I - B^-1) D^-1/2, full symmetric, from the lower triangle of B^-1
__global__ __launch_bounds__(256)
void k_sigma(const double* __restrict__ Binv, const double* __restrict__ s, int N, int ld,
             double* __restrict__ out)
{
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= N) return;
    const int hi = m > n ? m : n, lo = m > n ? n : m;
    const double b = Binv[(size_t)hi * ld + lo];
    out[(size_t)m * ld + n] = ((m == n ? 1.0 : 0.0) - b) / (s[m] * s[n]);
}

int vec_sigma(gprn_ctx* c, const double* Binv, const double* s, double* out)
{
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_sigma, dim3((c->N + 255) / 256, c->N), dim3(256), 0, c->stream, Binv, s,
                       c->N, c->ld, out);
    LAUNCH_END(c);
}

// ---- gradient contraction: per row m of G = 1/2 (P - Kinv + a a^T),
//   part[m][l] = sum_n G[m][n] dK[m][n]/dtheta_l,   l < 4,
// for the three kernels with closed forms here (SE: theta, ell; Periodic: theta, P, ell; QP: theta, le, P, lp --
// the formulas of covfunc._dk_dpars), one wave per row, rows summed by k_sum_cols.  a = Kinv m is formed first.
__global__ __launch_bounds__(256)
void k_symv(const double* __restrict__ M, const double* __restrict__ v, int N, int ld, double* __restrict__ out)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    double acc = 0.0;
    for (int c = lane; c < N; c += 64) acc += M[(size_t)i * ld + c] * v[c];
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
}

__global__ __launch_bounds__(256)
void k_grad_rows(int kid, double q0, double q1, double q2, double q3, const double* __restrict__ t,
                 const double* __restrict__ Kinv, const double* __restrict__ P, const double* __restrict__ a,
                 int N, int ld, double* __restrict__ part /* [N][4] */)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= N) return;
    const double tm = t[m], am = a[m];
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
    for (int n = lane; n < N; n += 64) {
        const double G = 0.5 * (P[(size_t)m * ld + n] - Kinv[(size_t)m * ld + n] + am * a[n]);
        const double r = tm - t[n];
        if (kid == GPRN_K_SE) {
            const double K = q0 * q0 * exp(-0.5 * (r * r) / (q1 * q1));
            g0 += G * (2 * K / q0);
            g1 += G * (K * (r * r) / (q1 * q1 * q1));
        } else if (kid == GPRN_K_PERIODIC) {
            const double x = 3.141592653589793 * fabs(r) / q1, sx = sin(x);
            const double K = q0 * q0 * exp(-2 * (sx * sx) / (q2 * q2));
            g0 += G * (2 * K / q0);
            g1 += G * (K * 2 * x * sin(2 * x) / (q1 * (q2 * q2)));
            g2 += G * (K * 4 * (sx * sx) / (q2 * q2 * q2));
        } else {                                   // GPRN_K_QP
            const double x = 3.141592653589793 * fabs(r) / q2, sx = sin(x);
            const double K = q0 * q0 * exp(-2 * (sx * sx) / (q3 * q3) - (r * r) / (2 * (q1 * q1)));
            g0 += G * (2 * K / q0);
            g1 += G * (K * (r * r) / (q1 * q1 * q1));
            g2 += G * (K * 2 * x * sin(2 * x) / (q2 * (q3 * q3)));
            g3 += G * (K * 4 * (sx * sx) / (q3 * q3 * q3));
        }
    }
    g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2); g3 = wave_sum(g3);
    if (lane == 0) { part[4 * m] = g0; part[4 * m + 1] = g1; part[4 * m + 2] = g2; part[4 * m + 3] = g3; }
}

__global__ __launch_bounds__(256)
void k_sum_cols4(const double* __restrict__ part, int n, double* __restrict__ out /* 4 */)
{
    __shared__ double sh[4];
    for (int l = 0; l < 4; ++l) {                  // fixed order: the result does not depend on the launch
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) acc += part[4 * i + l];
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) out[l] = acc;
        __syncthreads();
    }
}

int vec_symv(gprn_ctx* c, const double* M, const double* v, double* out)
{
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_symv, dim3((c->N + 3) / 4), dim3(256), 0, c->stream, M, v, c->N, c->ld, out);
    LAUNCH_END(c);
}

int vec_grad_contract(gprn_ctx* c, int kid, const double* par, const double* Kinv, const double* P, const double* m,
                      double* a_scratch, double* part_scratch, double* out4)
{
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_symv, dim3((c->N + 3) / 4), dim3(256), 0, c->stream, Kinv, m, c->N, c->ld, a_scratch);
    hipLaunchKernelGGL(k_grad_rows, dim3((c->N + 3) / 4), dim3(256), 0, c->stream, kid, par[0], par[1], par[2], par[3],
                       c->d_time, Kinv, P, a_scratch, c->N, c->ld, part_scratch);
    hipLaunchKernelGGL(k_sum_cols4, dim3(1), dim3(256), 0, c->stream, part_scratch, c->N, out4);
    LAUNCH_END(c);
}

// dst += src on the N x N block of two ld-pitched matrices (the node sum of quirk Q1 for the gradient)
__global__ __launch_bounds__(256)
void k_axpy_matrix(const double* __restrict__ src, double* __restrict__ dst, int N, int ld)
{
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n < N && m < N) dst[(size_t)m * ld + n] += src[(size_t)m * ld + n];
}

int vec_axpy_matrix(gprn_ctx* c, const double* src, double* dst, int N)
{
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_axpy_matrix, dim3((N + 255) / 256, N), dim3(256), 0, c->stream, src, dst, N, c->ld);
    LAUNCH_END(c);
}

// upper triangle := transpose of the lower one (ld x ld)
__global__ __launch_bounds__(256)
void k_symmetrize(double* __restrict__ M, int ld)
{
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n < ld && n > m) M[(size_t)m * ld + n] = M[(size_t)n * ld + m];
}

int vec_symmetrize(gprn_ctx* c, double* M)
{
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_symmetrize, dim3((c->ld + 255) / 256, c->ld), dim3(256), 0, c->stream, M, c->ld);
    LAUNCH_END(c);
}

// ---- prediction (gprn_predict): mean[i] = sum_n Ks[i][n] sol[n];  q[i] = sum_a WT[i][a]^2
__global__ __launch_bounds__(256)
void k_pred_rows(double* const* __restrict__ ptrs, int ns, int N, int ld, int ns_pad,
                 const double* __restrict__ sol, const double* __restrict__ kss,
                 double* __restrict__ mean, double* __restrict__ var)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= ns) return;
    const double* Ks = ptrs[(size_t)slot * GPRN_NBUF + BUF_K] + (size_t)i * ld;
    const double* WT = ptrs[(size_t)slot * GPRN_NBUF + BUF_KLINV] + (size_t)i * ld;
    const double* s = sol + (size_t)slot * ld;
    double m = 0.0, q = 0.0;
    for (int n = lane; n < N; n += 64) {
        m += Ks[n] * s[n];
        const double w = WT[n];
        q += w * w;
    }
    m = wave_sum(m);
    q = wave_sum(q);
    if (lane == 0) {
        mean[(size_t)slot * ns_pad + i] = m;
        var[(size_t)slot * ns_pad + i] = kss[(size_t)slot * ns_pad + i] - q;
    }
}

int vec_pred_rows(gprn_ctx* c, int nslots, int ns, int ns_pad, const double* sol, const double* kss,
                  double* mean, double* var)
{
    if (!nslots) return GPRN_OK;
    prof_begin(c, GPRN_T_VEC);
    hipLaunchKernelGGL(k_pred_rows, dim3((ns + 3) / 4, nslots), dim3(256), 0, c->stream,
                       (double* const*)c->d_ptrs, ns, c->N, c->ld, ns_pad, sol, kss, mean, var);
    LAUNCH_END(c);
}
