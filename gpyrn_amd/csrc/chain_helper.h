// The chain's two products of a tile step INSIDE the diagonal-block launch (round 4; device code, included by
// factor.hip and the probe _probe/step_bench.hip).
//
// Tile step k of the factorisation (factor.hip) is, on the chain,
//     diag(k):  L_kk = chol(B_kk), X_kk = L_kk^-1      ->   L_{k+1,k} = B_{k+1,k} L_kk^-T   ->   B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T
// and until round 4 those were three launches: 23.5 + 2.9 + 2.3 us of kernels and 8.8 us of launch boundaries per step
// at BASELINE config 2 (profiles/r03_chain_stamps_c2.txt).  Here the second and third run as ONE helper workgroup per
// matrix in the SAME launch as the diagonal block, on a CU of its own, beside it in time: the diagonal-block kernel
// publishes column block j of L_kk and the 16 x 16 inverse X_jj when its phase j ends (diag_tile.h, DiagPub), and the
// helper does the substitution column block by column block as they arrive,
//     L[:, j]  = B~[:, j] X_jj^T                        (i)    B~ = B_{k+1,k} with the columns before j eliminated
//     B~[:, m] -= L[:, j] L_kk[m, j]^T,  m > j          (ii)
//     U[P, Q] -= L[P, j] L[Q, j]^T,      P >= Q         (iii)  the lower 16 x 16 blocks of B_{k+1,k+1}
// so that the work of a phase SHRINKS towards the end (100, 92, ... 44 MFMAs per wave): what is left when the
// diagonal block retires is (i) + (iii) of the last column block.  (The product form L = B X_kk^T, which the separate
// launches use, needs row block j of X_kk, whose K grows with j: 256 of its 1152 MFMAs would sit behind the last phase.)
// This replaces, for the tile below the diagonal, jax's triangular solve inside the reference's Cholesky
// (/root/reference/gpyrn/meanfield.py:71-89); rounding differs from the product form at the 1e-16 level.
//
// Layout.  Four waves, one per SIMD (the launch is the diagonal block's: one wave per SIMD, 512 VGPRs each); wave W owns
// row blocks W and 7 - W of the tile (9 of the 36 U blocks each).  Everything a wave multiplies lives in registers in
// MFMA layouts and is never transposed:
//     bt[a][m]   (B~[R_a, m])^T as an accumulator: lane (fr, fk), register t  =  B~[16 R_a + fr][16 m + fk + 4 t]
//                -- which is ALSO the B operand of (i) (K index fk + 4 t in the t-th of four K = 4 products);
//     lt[a]      (L[R_a, j])^T out of (i), same layout: the B operand of (ii) and both operands of (iii);
//     X_jj, L_kk[m, j]   A operands, lane (fr, fk) loads element [fr][fk + 4 t] straight from memory (agent scope);
//     u[a][Q]    U[R_a, Q] as an accumulator, lane (fr, fk), register t = U[16 R_a + fk + 4 t][16 Q + fr].
// The other waves' L blocks for (iii) travel through LDS register image by register image ([block][t][lane]): one
// workgroup barrier per phase, two buffers by parity.
#pragma once
#include "gprn_internal.h"
#include "diag_tile.h"

#define HX_BLOCK (4 * 64)                       // doubles of one exchanged block: [t][lane]
#define HX_DOUBLES (2 * NSB * HX_BLOCK)         // two parities x 8 row blocks: 32 KiB (the diagonal block's LDS is 46.6)

// *flag has reached `target` (sequence numbers that wrap: the difference is what counts); bounded like spin_until
__device__ __forceinline__ void spin_until_seq(const unsigned* flag, unsigned target, unsigned* timed_out)
{
    if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long budget = timed_out ? (unsigned long long)timed_out[1] : 200000000ull;
        for (;;) {
            __builtin_amdgcn_s_sleep(2);
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

__device__ __forceinline__ double ld_agent(gcptr_t p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef STEP_STAMPS        // _probe/step_bench.hip: 100 MHz stamps of the helper's wave 0
__device__ unsigned long long helper_stamps[NSB + 2][2];
#define HP_STAMP(j, i) do { if (threadIdx.x == 0) helper_stamps[j][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HP_STAMP(j, i) do {} while (0)
#endif

// phase J of wave W (both compile-time: every register-array index is a constant)
template <int W, int J>
__device__ __forceinline__ void helper_phase(v4d (&bt)[2][NSB], v4d (&u)[2][NSB], double* __restrict__ ex, gcptr_t Lkk,
                                             gcptr_t Xkk, gptr_t Bl, int ld, const unsigned* prog, unsigned base,
                                             unsigned* timed_out)
{
    constexpr int R[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    if (lane == 0) spin_until_seq(prog, base + J + 1, timed_out);
    asm volatile("" ::: "memory");
    HP_STAMP(J, 0);
    // operands from the diagonal-block kernel's stores of this phase (agent scope on both sides)
    double xa[4], la[NSB][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xa[t] = ld_agent(Xkk + (size_t)(16 * J + fr) * ld + 16 * J + fk + 4 * t);
#pragma unroll
    for (int m = 0; m < NSB; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (m > J) la[m][t] = ld_agent(Lkk + (size_t)(16 * m + fr) * ld + 16 * J + fk + 4 * t);
    // (i)  (L[R_a, J])^T = X_JJ (B~[R_a, J])^T
    v4d lt[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) lt[a] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) lt[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[t], bt[a][J][t], lt[a], 0, 0, 0);
    double* const exj = ex + (J & 1) * NSB * HX_BLOCK;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            exj[R[a] * HX_BLOCK + t * 64 + lane] = lt[a][t];
            Bl[(size_t)(16 * R[a] + fr) * ld + 16 * J + fk + 4 * t] = lt[a][t];      // L_{k+1,k}, in place
        }
    // (ii)  (B~[R_a, m])^T -= L_kk[m, J] (L[R_a, J])^T
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < NSB; ++m)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (m > J) bt[a][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[m][t], lt[a][t], bt[a][m], 0, 0, 0);
    __syncthreads();                               // every row block of column J is in LDS
    // (iii)  U[R_a, Q] -= L[R_a, J] L[Q, J]^T
    double lq[NSB][4];
#pragma unroll
    for (int Q = 0; Q < NSB; ++Q)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (Q <= R[0] || Q <= R[1]) lq[Q][t] = exj[Q * HX_BLOCK + t * 64 + lane];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int Q = 0; Q < NSB; ++Q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (Q <= R[a]) u[a][Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lt[a][t], lq[Q][t], u[a][Q], 0, 0, 0);
    HP_STAMP(J, 1);
}

template <int W>
__device__ __forceinline__ void helper_wave(double* __restrict__ ex, gcptr_t Lkk, gcptr_t Xkk, gptr_t Bl, gptr_t Bu, int ld,
                                            const unsigned* prog, unsigned base, unsigned* timed_out)
{
    constexpr int R[2] = {W, 7 - W};
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    v4d bt[2][NSB], u[2][NSB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < NSB; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bt[a][m][t] = Bl[(size_t)(16 * R[a] + fr) * ld + 16 * m + fk + 4 * t];
                u[a][m][t] = (m <= R[a]) ? Bu[(size_t)(16 * R[a] + fk + 4 * t) * ld + 16 * m + fr] : 0.0;
            }
    helper_phase<W, 0>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 1>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 2>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 3>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 4>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 5>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 6>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
    helper_phase<W, 7>(bt, u, ex, Lkk, Xkk, Bl, ld, prog, base, timed_out);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int Q = 0; Q < NSB; ++Q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (Q <= R[a]) Bu[(size_t)(16 * R[a] + fk + 4 * t) * ld + 16 * Q + fr] = u[a][Q][t];
}

// All 256 threads of the helper workgroup call it.  Lkk / Xkk: tile (k, k) of B (L) and of X as the diagonal-block
// workgroup of the same launch writes them; Bl: tile (k+1, k) of B, overwritten with L_{k+1,k}; Bu: tile (k+1, k+1),
// lower blocks updated.  `lds`: HX_DOUBLES doubles.  The caller has waited for whatever wrote Bl and Bu.
__device__ __forceinline__ void chain_helper(double* __restrict__ lds, gcptr_t Lkk, gcptr_t Xkk, gptr_t Bl, gptr_t Bu, int ld,
                                             const unsigned* prog, unsigned base, unsigned* timed_out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) helper_wave<0>(lds, Lkk, Xkk, Bl, Bu, ld, prog, base, timed_out);
    else if (wave == 1) helper_wave<1>(lds, Lkk, Xkk, Bl, Bu, ld, prog, base, timed_out);
    else if (wave == 2) helper_wave<2>(lds, Lkk, Xkk, Bl, Bu, ld, prog, base, timed_out);
    else helper_wave<3>(lds, Lkk, Xkk, Bl, Bu, ld, prog, base, timed_out);
}
