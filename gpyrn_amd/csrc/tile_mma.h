// The tile contraction of k_tile_gemm / k_tile_panel (gemm_tile.hip): one workgroup of WM x WN waves computes a
// BM x BN tile  C (op)= A . B  over klen on v_mfma_f64_16x16x4_f64.  See gemm_tile.hip for the design notes.
#pragma once
#include "gprn_internal.h"

#include <type_traits>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#ifndef GPRN_DEEP_PREFETCH
#define GPRN_DEEP_PREFETCH 1
#endif

// Staging geometry.  A thread moves R*8/NT 16-byte pieces of an operand chunk (R rows x 16 k); piece `it`
// differs from piece 0 by +64 rows (only when the lanes cover 64 of 128 rows) and/or +8 k in BOTH memory
// layouts, so one lane offset serves all pieces: memory address = base + lane offset + a uniform per-piece
// offset (the buffer load's scalar offset), LDS address = lane base + an immediate.
//   mode 0, element (row, k) at row*ld + k: piece = (row, 2kp), (row, 2kp+1); lane = (kp & 3, row)
//   mode 1, element (row, k) at k*ld + row: piece = (2rp, k), (2rp+1, k);     lane = (rp, k & 7)
template <int R, int NT>
__device__ __forceinline__ void lane_geometry(int tid, int mode, int ld, unsigned& goff, int& l0)
{
    constexpr int P = R + 16;
    constexpr int RL = (R < NT / 4) ? R : NT / 4;          // rows the lane index spans (64 or 128)
    static_assert(R * 8 >= NT && (RL == 64 || RL == 128), "unsupported tile / workgroup combination");
    if (mode == 0) {
        const int kp = tid & 3, row = (tid >> 2) & (RL - 1);
        goff = ((unsigned)row * (unsigned)ld + 2u * kp) * 8u;
        l0 = (2 * kp) * P + (row ^ (4 * kp));
    } else {
        const int rp = tid & (RL / 2 - 1), k = tid / (RL / 2);      // k < 8
        goff = ((unsigned)k * (unsigned)ld + 2u * rp) * 8u;
        l0 = k * P + ((2 * rp) ^ (4 * ((k >> 1) & 3)));
    }
}

// lds: 2 * 16 * (BM + BN + 32) doubles, 16-byte aligned.  A, B, C point at this workgroup's part of the
// operands; modes as in TileTask; mb16_0 / nb16_0: index of the part's first 16-row / 16-column block inside
// its 128 x 128 tile (TRI).  Every thread of the workgroup calls it; the last LDS reads are retired on return
// only after the caller's next barrier.
// LOWER: the part lies on the diagonal of a symmetric update C -= A A^T of which only the lower triangle is ever read
// (the factorisation's diagonal tiles): 16 x 16 blocks strictly above the diagonal (block column > block row inside the
// 128 x 128 tile) are loaded and stored back unchanged, their MFMAs skipped.
// ft_K (first touch of a tile of B = I + D^1/2 K D^1/2, CM_SUB only): the incoming C values are not read from C but
// formed from K (same offsets as C) and s = sqrt(d): delta + (s_row s_col) K, zero outside the n x n problem -- what
// k_build_B would have written there (same expression, same rounding).  ft_row / ft_col: the part's first row / column
// inside the matrix.
// DEEP_OK: the caller's register budget allows the second set of staging registers (the 64 x 64 form: 110 registers per lane)
// SYM: the part lies on the diagonal of a symmetric update of a diagonal tile, B_jj -= L[j,.] L[j,.]^T (task bit 4).  Its
// 16 x 16 blocks ON the diagonal hold the matrix' diagonal entries, which are ~1 in B = I + D^1/2 K D^1/2 where the update
// is ~d K: with the tile in the accumulator from the start (acc = -C, acc += A.B) every one of the klen / 4 MFMA steps
// rounds at the magnitude of C, and the errors do not average out -- 40 ulp on a pivot after 950 columns, which
// var = (1 - sum_r X_rc^2) / d divides by d ~ 2e-4 (profiles/r05_var_accuracy.txt: the device's variances were 4-11 x
// less accurate than LAPACK's on the same algebra).  Those blocks accumulate A.B from ZERO and the tile comes in at the
// end, C - acc: one rounding at C's magnitude per update, as a BLAS syrk does it.  (The other blocks hold off-diagonal
// entries, as small as their updates; they keep the form that needs no second pass over C.)
template <int BM, int BN, int WM, int WN, int TRI, bool LOWER = false, bool DEEP_OK = false, bool SYM = LOWER>
__device__ __forceinline__ void tile_mma(double* lds, const double* A, const double* B, gptr_t C, int ld,
                                         int a_mode, int b_mode, int c_mode, int klen, int mb16_0, int nb16_0,
                                         const double* ft_K = nullptr,
                                         const double* ft_s = nullptr, int ft_row = 0, int ft_col = 0, int ft_n = 0)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: conditions on it are scalar branches
    const int wr = wave / WN, wc = wave % WN;
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;                   // a wave's part of the tile
    constexpr int MI = TM / 16, NI = TN / 16;                   // 16x16 MFMA tiles per wave
    constexpr int PA = BM + 16, PB = BN + 16;                   // LDS pitches, doubles
    constexpr int A_DOUBLES = 16 * PA, B_DOUBLES = 16 * PB;
    constexpr int STAGE = A_DOUBLES + B_DOUBLES;
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;       // 16-byte loads per thread per chunk
    constexpr bool A_ROWS2 = BM * 4 > NT, B_ROWS2 = BN * 4 > NT;   // the pieces of a thread span 2 x 64 rows
    static_assert(A_IT >= 1 && B_IT >= 1 && NI >= 1 && (MI % 2 == 0) && (NI % 2 == 0), "tile too small for the workgroup");

    // ---- global -> LDS staging (lane_geometry): buffer loads = uniform base (advanced per chunk) + lane
    // offset + uniform piece offset; the two doubles of a piece go to lane base (+ dl) + immediate
    const size_t a_step = a_mode ? (size_t)GPRN_KC * ld : (size_t)GPRN_KC;       // doubles per chunk
    const size_t b_step = b_mode ? (size_t)GPRN_KC * ld : (size_t)GPRN_KC;
    unsigned a_g, b_g;
    int a_l, b_l;
    lane_geometry<BM, 64 * WM * WN>(tid, a_mode, ld, a_g, a_l);
    lane_geometry<BN, 64 * WM * WN>(tid, b_mode, ld, b_g, b_l);
    const int a_l0 = a_l * 8, a_l1 = a_l0 + (a_mode ? 1 : PA) * 8;              // LDS byte addresses, stage 0
    const int b_l0 = (A_DOUBLES + b_l) * 8, b_l1 = b_l0 + (b_mode ? 1 : PB) * 8;
    // uniform piece offsets in memory, bytes: +64 rows, +8 k
    const unsigned a_row64 = (a_mode ? 64u : 64u * (unsigned)ld) * 8u, a_k8 = (a_mode ? 8u * (unsigned)ld : 8u) * 8u;
    const unsigned b_row64 = (b_mode ? 64u : 64u * (unsigned)ld) * 8u, b_k8 = (b_mode ? 8u * (unsigned)ld : 8u) * 8u;
    // ---- MFMA operand fetch: lane holds A[row = fr][k = fk], B[k = fk][col = fr]; LDS byte address of
    // k4-step ks = lane base[ks] + immediate (16-row block, stage)
    const int fr = lane & 15, fk = lane >> 4;
    const int mb16 = mb16_0 + ((wr * TM) >> 4), nb16 = nb16_0 + ((wc * TN) >> 4);   // wave's first 16-blocks
    const int row0 = wr * TM;                                  // first row of the wave's block i = 0
    int a_fb[4], b_fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int k = 4 * ks + fk, sw = fr ^ (4 * ((k >> 1) & 3));
        a_fb[ks] = (k * PA + row0 + sw) * 8;
        b_fb[ks] = (A_DOUBLES + k * PB + wc * TN + sw) * 8;
    }
    const char* const lds_b = reinterpret_cast<const char*>(lds);
    char* const lds_w = reinterpret_cast<char*>(lds);
    auto frag = [&](int byte_addr) { return *reinterpret_cast<const double*>(lds_b + byte_addr); };

    const bool neg = c_mode != CM_SET;                       // acc holds -(result)
    auto on_diag = [&](int i, int j) { return SYM && nb16 + j == mb16 + i; };   // (wave-uniform)
    gptr_t Cw = C + (size_t)(row0 + fk) * ld + wc * TN + fr;
    auto crow = [&](int i) { return (size_t)(i * 16); };       // rows of block i past Cw
    auto arow_bytes = [&](int i) { return i * 128; };          // same, LDS bytes
    v4d acc[MI][NI];

    const int nchunks = klen / GPRN_KC;
    // DEEP (64 x 64 on four waves): global loads THREE chunks ahead through two register sets (set A holds odd chunks, B
    // even ones from chunk 2 on) instead of two chunks ahead through one.  PMC over the bulk launches: a wave spends 60 %
    // of its cycles at a wait, a third of the L2 requests miss, and with two or three waves per SIMD that leaves the
    // matrix pipes idle a third of the time -- one chunk period (16 MFMAs per wave) is not enough to cover an L2 miss.
    // (GPRN_DEEP_PREFETCH=2: the 4-wave 64 x 128 / 128 x 64 forms too, i.e. the panel products)
    constexpr bool DEEP = DEEP_OK && WM == 2 && WN == 2 &&
                          ((GPRN_DEEP_PREFETCH >= 1 && BM == 64 && BN == 64) || (GPRN_DEEP_PREFETCH >= 2 && BM * BN == 64 * 128));
    v2d ra[A_IT], rb[B_IT];
    v2d ra2[DEEP ? A_IT : 1], rb2[DEEP ? B_IT : 1];
    auto load_chunk = [&](v2d (&ra)[A_IT], v2d (&rb)[B_IT]) {
        // raw buffer resources over the chunk's base: 48-bit address, no stride, no bounds (num_records max)
        const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const unsigned so = A_ROWS2 ? (it & 1) * a_row64 + (it >> 1) * a_k8 : it * a_k8;
            ra[it] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, a_g, so, 0));
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const unsigned so = B_ROWS2 ? (it & 1) * b_row64 + (it >> 1) * b_k8 : it * b_k8;
            rb[it] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, b_g, so, 0));
        }
    };
    auto write_chunk = [&](int stage_bytes, const v2d (&ra)[A_IT], const v2d (&rb)[B_IT]) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int imm = stage_bytes + (A_ROWS2 ? (it & 1) * 64 * 8 + (it >> 1) * 8 * PA * 8 : it * 8 * PA * 8);
            *reinterpret_cast<double*>(lds_w + a_l0 + imm) = ra[it][0];
            *reinterpret_cast<double*>(lds_w + a_l1 + imm) = ra[it][1];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int imm = stage_bytes + (B_ROWS2 ? (it & 1) * 64 * 8 + (it >> 1) * 8 * PB * 8 : it * 8 * PB * 8);
            *reinterpret_cast<double*>(lds_w + b_l0 + imm) = rb[it][0];
            *reinterpret_cast<double*>(lds_w + b_l1 + imm) = rb[it][1];
        }
    };

    load_chunk(ra, rb);
    // The C tile is requested AFTER the first operand chunk: memory returns in order, so the
    // LDS staging below waits only for the chunk, and the first MFMA of each accumulator only for
    // its own four values -- most of the tile streams in behind the first MFMAs.
    __builtin_amdgcn_sched_barrier(0);
    if (c_mode == CM_SUB && ft_K) {              // first touch: the tile of B is built here
        gcptr_t Kw = (gcptr_t)ft_K + (size_t)(row0 + fk) * ld + wc * TN + fr;
        double sc[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) sc[j] = ft_s[ft_col + wc * TN + fr + 16 * j];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = ft_row + row0 + fk + (int)crow(i) + 4 * r;
                const double sm = m < ft_n ? ft_s[m] : 0.0;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = ft_col + wc * TN + fr + 16 * j;
                    const double kv = (m < ft_n && n < ft_n) ? Kw[(crow(i) + 4 * r) * ld + j * 16] : 0.0;
                    double v = (n < ft_n) ? sm * sc[j] * kv : 0.0;
                    if (m == n) v += 1.0;
                    acc[i][j][r] = on_diag(i, j) ? 0.0 : -v;       // (SYM: formed again in the epilogue)
                }
            }
    } else if (c_mode == CM_SUB) {               // one uniform branch around all the loads
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (on_diag(i, j)) { acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0}; continue; }   // (SYM: read in the epilogue)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = -Cw[(crow(i) + 4 * r) * ld + j * 16];
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    __builtin_amdgcn_sched_barrier(0);
    write_chunk(0, ra, rb);
    if (nchunks > 1) { A += a_step; B += b_step; }
    load_chunk(ra, rb);                          // chunk 1 (chunk 0 again when there is only one)
    if constexpr (DEEP) {
        if (nchunks > 2) { A += a_step; B += b_step; }
        load_chunk(ra2, rb2);                    // chunk 2 (or the last one once more)
    }
    __syncthreads();

    // fragments of the k4-step about to be multiplied: B's are fetched a whole step ahead (every
    // MFMA of a step reads them), A's row block by row block as the previous step lets go of them
    double af[MI], bf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = frag(a_fb[0] + arow_bytes(i));
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[0][j] = frag(b_fb[0] + j * 128);

    // One K-chunk: four k4-steps of MI*NI MFMAs.  Beside them: the fragments of the next step, the
    // staging of chunk c+1 into the other LDS stage (second step; its global loads were issued a
    // chunk ago), the request of chunk c+2, and -- before the last step -- the one barrier that
    // publishes stage c+1 and retires the reads of stage c, so that the last step can already fetch
    // the first fragments of the next chunk.  The sched_group_barrier sequences spread the memory
    // instructions between the MFMAs (each holds the matrix pipe for 64 cycles: whatever issues in
    // its shadow is free, whatever is clustered between two of them is not).
    auto chunk = [&](auto last_c, int sb, int c, v2d (&rx)[A_IT], v2d (&ry)[B_IT]) {
        constexpr bool LAST = decltype(last_c)::value;
        const int nb = sb ^ (STAGE * 8);                       // byte offsets of this chunk's and the next one's stage
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const bool fetch = ks < 3 || !LAST;                // there is a next step
            const int fb_off = ks < 3 ? sb : nb, fb_ks = ks < 3 ? ks + 1 : 0;
            if (fetch) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[nxt][j] = frag(fb_off + b_fb[fb_ks] + j * 128);
            }
            if (ks == 1 && !LAST) {
                write_chunk(nb, rx, ry);
                // chunk c+2 (DEEP: c+3), or once more the last one (its registers are not read again)
                const bool more = c + (DEEP ? 3 : 2) < nchunks;
                A += more ? a_step : 0;
                B += more ? b_step : 0;
                load_chunk(rx, ry);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (TRI == 1 && c > nb16 + j) continue;     // wave-uniform
                    if (TRI == 2 && c > mb16 + i) continue;
                    if (LOWER && nb16 + j > mb16 + i) continue;   // wave-uniform
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[cur][j], acc[i][j], 0, 0, 0);
                }
                if (fetch) af[i] = frag(fb_off + a_fb[fb_ks] + arow_bytes(i));
            }
            if (TRI == 0) {
                // issue order of this step: MFMAs and memory instructions in turn (adjacent fragment
                // reads pair up into ds_read2_b64: (MI + NI) / 2 read instructions per step)
                constexpr int NMF = MI * NI, NRD = (MI + NI) / 2;
                constexpr int NWR = 2 * (A_IT + B_IT), NLD = A_IT + B_IT;
                if (ks == 1 && !LAST) {
#pragma unroll
                    for (int g = 0; g < NLD; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NMF / NLD, 0);
                        if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, NWR / NLD, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                } else if (fetch) {
#pragma unroll
                    for (int g = 0; g < NRD; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NMF / NRD, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
            if (ks == 2 && !LAST) __syncthreads();
        }
    };
    int sb = 0;
    if constexpr (DEEP) {
        // chunk c writes chunk c+1 to LDS from the set that holds it (odd: A, even: B) and refills that set with c+3
        int c = 0;
        for (; c + 1 < nchunks - 1; c += 2) {
            chunk(std::false_type{}, sb, c, ra, rb);
            sb ^= STAGE * 8;
            chunk(std::false_type{}, sb, c + 1, ra2, rb2);
            sb ^= STAGE * 8;
        }
        if (c < nchunks - 1) {
            chunk(std::false_type{}, sb, c, ra, rb);
            sb ^= STAGE * 8;
        }
        chunk(std::true_type{}, sb, nchunks - 1, ra, rb);
    } else {
        for (int c = 0; c < nchunks - 1; ++c) {
            chunk(std::false_type{}, sb, c, ra, rb);
            sb ^= STAGE * 8;
        }
        chunk(std::true_type{}, sb, nchunks - 1, ra, rb);
    }

    // ---- epilogue: C/D layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (SYM && c_mode == CM_SUB && on_diag(i, j)) {
                // the block accumulated A.B from zero: the tile comes in now (first touch: formed from K as above)
                double cin[4];
                if (ft_K) {
                    gcptr_t Kw = (gcptr_t)ft_K + (size_t)(row0 + fk) * ld + wc * TN + fr;
                    const int n = ft_col + wc * TN + fr + 16 * j;
                    const double scj = n < ft_n ? ft_s[n] : 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = ft_row + row0 + fk + (int)crow(i) + 4 * r;
                        const double sm = m < ft_n ? ft_s[m] : 0.0;
                        const double kv = (m < ft_n && n < ft_n) ? Kw[(crow(i) + 4 * r) * ld + j * 16] : 0.0;
                        double v = (n < ft_n) ? sm * scj * kv : 0.0;
                        if (m == n) v += 1.0;
                        cin[r] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cin[r] = Cw[(crow(i) + 4 * r) * ld + j * 16];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Cw[(crow(i) + 4 * r) * ld + j * 16] = cin[r] - acc[i][j][r];
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Cw[(crow(i) + 4 * r) * ld + j * 16] = neg ? -acc[i][j][r] : acc[i][j][r];
            }
        }
}
