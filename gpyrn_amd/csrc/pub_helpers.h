// Consumers of a diagonal block that is still being factored (round 4; device code, included by factor.hip).
//
// The diagonal-block kernel (diag_tile.h) publishes its results phase by phase when it is given a DiagPub: after phase j
// (j = 0..7) column block j of L_kk, the 16 x 16 inverse X_jj and row block j of X_kk = L_kk^-1 are in memory and the
// phase counter stands at base + j + 1.  A tile step's PANEL -- the products of the other tiles of column k / row k with
// the diagonal tile, factor.hip -- does not have to wait for the whole block: a workgroup per tile, launched on the side
// stream with its operand already in registers, follows the counter and is done a phase's work after the diagonal
// block instead of a 64 x 128 x 128 product's pipeline later:
//   below the diagonal   L_ik = B_ik L_kk^-T  by substitution, column block by column block (trsm_tile):
//                            L[:, j]  = B~[:, j] X_jj^T                       (i)
//                            B~[:, m] -= L[:, j] L_kk[m, j]^T,  m > j         (ii)   -- the work per phase SHRINKS with j
//   left of it           X_kc = X_kk R_kc     row block by row block (xmul_tile):
//                            X_kc[j, :] = sum_{m <= j} X_kk[j, m] R_kc[m, :]         -- grows with j: 64 MFMAs per wave at the end
// This is jax's triangular solve inside the reference's Cholesky resp. its cho_solve
// (/root/reference/gpyrn/meanfield.py:71-89, 1041): (i)/(ii) round differently from the product form L = B X_kk^T that
// k_tile_panel computes (1e-16 relative).
//
// Layouts.  Four waves, wave W owns row blocks (trsm) / column blocks (xmul) W and 7 - W of the 128 x 128 tile; what a wave
// multiplies lives in registers in MFMA layouts and is never transposed (lane = (fr, fk), fr = lane & 15, fk = lane >> 4;
// the t-th of the four K = 4 products of a 16 x 16 x 16 block product takes K index fk + 4 t on both sides):
//   trsm   bt[a][m]  (B~[R_a, m])^T as an accumulator: register t = B~[16 R_a + fr][16 m + fk + 4 t] -- which IS the B
//                    operand of (i); its result lt[a] = (L[R_a, j])^T has the same layout and is the B operand of (ii);
//                    X_jj and L_kk[m, j] are A operands, element [fr][fk + 4 t], loaded at agent scope.
//   xmul   rb[b][m]  R[m][:, C_b] as B operand: register t = R[16 m + fk + 4 t][16 C_b + fr]; X_kk[j, m] as A operand;
//                    the result in accumulator layout, register t = out[16 j + fk + 4 t][16 C_b + fr] (row-contiguous stores).
#pragma once
#include "gprn_internal.h"
#include "diag_tile.h"

// *flag has reached `target` (sequence numbers that wrap: the difference is what counts); bounded like spin_until
__device__ __forceinline__ void spin_until_seq(const unsigned* flag, unsigned target, unsigned* timed_out)
{
    if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long budget = timed_out ? (unsigned long long)timed_out[1] : 200000000ull;
        for (;;) {
            __builtin_amdgcn_s_sleep(4);
            if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
            if (timed_out && __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
                if (timed_out && atomicExch(timed_out, 1u) == 0u) timed_out[2] = (unsigned)(flag - timed_out);
                break;
            }
        }
    }
    asm volatile("" ::: "memory");
}

// what the diagonal-block kernel stored at agent scope (st_pub): read at agent scope, past this XCD's L2
__device__ __forceinline__ double ld_agent(gcptr_t p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- L_ik = B_ik L_kk^-T, in place over tile Bl; Lkk / Xkk: the diagonal tile of B (L) and of X as they are published
template <int W, int J>
__device__ __forceinline__ void trsm_phase(v4d (&bt)[2][NSB], gcptr_t Lkk, gcptr_t Xkk, gptr_t Bl, int ld,
                                           const unsigned* prog, unsigned base, unsigned* timed_out)
{
    constexpr int R[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    if (lane == 0) spin_until_seq(prog, base + J + 1, timed_out);
    asm volatile("" ::: "memory");
    double xa[4], la[NSB][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xa[t] = ld_agent(Xkk + (size_t)(16 * J + fr) * ld + 16 * J + fk + 4 * t);
#pragma unroll
    for (int m = 0; m < NSB; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (m > J) la[m][t] = ld_agent(Lkk + (size_t)(16 * m + fr) * ld + 16 * J + fk + 4 * t);
    v4d lt[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) lt[a] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) lt[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[t], bt[a][J][t], lt[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) Bl[(size_t)(16 * R[a] + fr) * ld + 16 * J + fk + 4 * t] = lt[a][t];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < NSB; ++m)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (m > J) bt[a][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[m][t], lt[a][t], bt[a][m], 0, 0, 0);
}

template <int W>
__device__ __forceinline__ void trsm_wave(gcptr_t Lkk, gcptr_t Xkk, gptr_t Bl, int ld, const unsigned* prog, unsigned base,
                                          unsigned* timed_out)
{
    constexpr int R[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    v4d bt[2][NSB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < NSB; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) bt[a][m][t] = Bl[(size_t)(16 * R[a] + fr) * ld + 16 * m + fk + 4 * t];
    trsm_phase<W, 0>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out); trsm_phase<W, 1>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    trsm_phase<W, 2>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out); trsm_phase<W, 3>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    trsm_phase<W, 4>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out); trsm_phase<W, 5>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    trsm_phase<W, 6>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out); trsm_phase<W, 7>(bt, Lkk, Xkk, Bl, ld, prog, base, timed_out);
}

// ---- X_kc = X_kk R_kc, in place over tile Xc (R_kc lives there); Xkk as it is published
template <int W, int J>
__device__ __forceinline__ void xmul_phase(const double (&rb)[2][NSB][4], gcptr_t Xkk, gptr_t Xc, int ld, const unsigned* prog,
                                           unsigned base, unsigned* timed_out)
{
    constexpr int C[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    if (lane == 0) spin_until_seq(prog, base + J + 1, timed_out);
    asm volatile("" ::: "memory");
    double xa[NSB][4];
#pragma unroll
    for (int m = 0; m < NSB; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (m <= J) xa[m][t] = ld_agent(Xkk + (size_t)(16 * J + fr) * ld + 16 * m + fk + 4 * t);
    v4d o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) o[b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < NSB; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (m <= J) o[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[m][t], rb[b][m][t], o[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) Xc[(size_t)(16 * J + fk + 4 * t) * ld + 16 * C[b] + fr] = o[b][t];
}

template <int W>
__device__ __forceinline__ void xmul_wave(gcptr_t Xkk, gptr_t Xc, int ld, const unsigned* prog, unsigned base, unsigned* timed_out)
{
    constexpr int C[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    double rb[2][NSB][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int m = 0; m < NSB; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) rb[b][m][t] = Xc[(size_t)(16 * m + fk + 4 * t) * ld + 16 * C[b] + fr];
    // every lane's loads have returned before the first store of this workgroup overwrites a row block (in place: the
    // four waves own disjoint COLUMN blocks, so only a wave's own stores touch what it loaded)
    xmul_phase<W, 0>(rb, Xkk, Xc, ld, prog, base, timed_out); xmul_phase<W, 1>(rb, Xkk, Xc, ld, prog, base, timed_out);
    xmul_phase<W, 2>(rb, Xkk, Xc, ld, prog, base, timed_out); xmul_phase<W, 3>(rb, Xkk, Xc, ld, prog, base, timed_out);
    xmul_phase<W, 4>(rb, Xkk, Xc, ld, prog, base, timed_out); xmul_phase<W, 5>(rb, Xkk, Xc, ld, prog, base, timed_out);
    xmul_phase<W, 6>(rb, Xkk, Xc, ld, prog, base, timed_out); xmul_phase<W, 7>(rb, Xkk, Xc, ld, prog, base, timed_out);
}

// All 256 threads of a workgroup call one of these (no LDS, no barrier: the waves are independent)
__device__ __forceinline__ void trsm_tile(gcptr_t Lkk, gcptr_t Xkk, gptr_t Bl, int ld, const unsigned* prog, unsigned base,
                                          unsigned* timed_out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) trsm_wave<0>(Lkk, Xkk, Bl, ld, prog, base, timed_out);
    else if (wave == 1) trsm_wave<1>(Lkk, Xkk, Bl, ld, prog, base, timed_out);
    else if (wave == 2) trsm_wave<2>(Lkk, Xkk, Bl, ld, prog, base, timed_out);
    else trsm_wave<3>(Lkk, Xkk, Bl, ld, prog, base, timed_out);
}
__device__ __forceinline__ void xmul_tile(gcptr_t Xkk, gptr_t Xc, int ld, const unsigned* prog, unsigned base, unsigned* timed_out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) xmul_wave<0>(Xkk, Xc, ld, prog, base, timed_out);
    else if (wave == 1) xmul_wave<1>(Xkk, Xc, ld, prog, base, timed_out);
    else if (wave == 2) xmul_wave<2>(Xkk, Xc, ld, prog, base, timed_out);
    else xmul_wave<3>(Xkk, Xc, ld, prog, base, timed_out);
}
