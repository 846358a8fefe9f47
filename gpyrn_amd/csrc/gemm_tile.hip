// Batched 128x128 tile contraction on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Every O(N^3) flop of the sweep goes through this one kernel: the trailing
// SYRK of the blocked Cholesky, the row updates that build L^-1 alongside it,
// the panel solves (as products with the inverted diagonal block) and the
// X^T X product behind quirk Q1 (see factor.hip for the task lists).
//
// Work decomposition: one 256-thread workgroup (4 waves, 2x2) per output tile of
// BM x BN in {64,128}^2; each wave a (BM/2)x(BN/2) sub-tile of 16x16 MFMA tiles
// (128x128: 16 independent accumulators = 128 VGPRs).  K is streamed in chunks of
// 16 through a double-buffered LDS image (global -> registers -> LDS, one barrier
// per chunk); global loads of chunk c+2 are in flight while chunk c is multiplied.
//
// LDS image: BOTH operands are kept [k][row] (pitch rows+16 doubles, rows XOR-swizzled
// by 4*((k>>1)&3)), whatever their layout in memory -- a k-contiguous operand is
// transposed on its way in (two ds_write_b64 per 16-byte load either way).  With one
// layout the fragment addresses of the inner loop are lane bases plus immediates:
// the loop body has no address arithmetic, only MFMA, ds_read, ds_write, global_load.
//   operand fetch (one f64 per lane, ds_read_b64): lanes (fr, fk) of a 32-lane half
//     hit 16*fk + (fr ^ s) mod 32 -- conflict-free (pitch = 16 mod 32);
//   k-contiguous staging: a 16-lane group writes rows 4a+rr (rr<4), pairs kp<4 to
//     (row ^ 4kp) -- 16 distinct; row-contiguous staging: 16 consecutive row pairs (2-way on
//     ds_write_b64's 32 banks, no more LDS cycles than the instruction takes to issue).
// 2 stages x 2 operands x 16 x (128+16) x 8 B = 72 KiB for 128x128: two workgroups per CU.
//
// C -= A.B runs as acc = -C, acc += A.B, C = -acc: the tile is read once, up front, with
// all loads back to back, the epilogue is stores only, and the operands need no sign.
#include "gprn_internal.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "tile_mma.h"
#include "diag_tile.h"

// Output tile of one workgroup: BM x BN in {64,128}^2, computed by NW = 4 waves (2 x 2) or 8 waves (2 x 4).
// A 128x128 task is cut into (128/BM) x (128/BN) workgroups (sub-tile index = blockIdx.x % that).
// 128x128 on 8 waves is the throughput shape: 64x32 per wave = 8 accumulators, under 128 VGPRs, so two
// workgroups = four waves per SIMD share a CU and cover each other's barriers and memory waits.  The
// smaller 4-wave shapes put a latency-bound launch -- a few dozen tasks with K = 128 on the factorisation's
// critical path -- on 2-4x as many CUs:
//   64x128 (rows split)  is safe for in-place panel tasks whose C tile is their A operand,
//   128x64 (cols split)  for in-place tasks whose C tile is their B operand,
//   64x64                for everything that is not in place.
// TRI: an operand of every task of the launch is the inverted diagonal block X_kk (lower triangular,
// explicit zeros above): 1 = it is B, transposed (panel product L_ik = B_ik X_kk^T: k <= n),
// 2 = it is A (X_kc = X_kk R_kc: k <= m).  K-chunks that only meet the zero half of a 16-wide
// block are skipped -- 28 of 64 block products.
// TAG only names the launch family in profiles (rocprofv3 reports one row per instantiation):
// TG_PANEL panel products, TG_INNER in-panel K=128 updates, TG_NEXT next-panel K=512 updates,
// TG_BULK bulk K=512 updates, TG_AHEAD their look-ahead part, TG_MISC the rest.
template <int BM, int BN, int NW, int TRI, int TAG>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2)
void k_tile_gemm(const TileTask* __restrict__ tasks, double* const* __restrict__ ptrs, int ld,
                 unsigned* sig_slot, unsigned sig_value, const unsigned* then_wait, unsigned then_value,
                 const unsigned* wait_flag, unsigned wait_value, unsigned* wait_timed_out, int xcd_map,
                 const double* ft_s, int ft_n,
                 unsigned* start_flag, unsigned start_value)
{
    // (gprn_ctx::start_flag_now: the flag of the launch before this one on the stream -- in memory once a workgroup of
    // this launch runs -- instead of a stream write, a 4.5 us kernel of its own, between the two)
    if (start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(start_flag, start_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    await_flag(wait_flag, wait_value, wait_timed_out);
    constexpr int WM = 2, WN = NW / 2;                          // waves: WM x WN
    constexpr int SM = GPRN_TILE / BM, SN = GPRN_TILE / BN;     // sub-tiles per task
    __shared__ __attribute__((aligned(16))) double lds[2 * 16 * (BM + BN + 32)];

    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in dispatch order, so
    // neighbours in the grid never share an L2.  Re-map in chunks of 16: the workgroup an XCD receives as its
    // c-th takes position c % 16 of chunk (c / 16) * 8 + x of the (task, sub-tile, matrix) list -- 16 consecutive
    // entries (the sub-tiles of a task, tasks that share the operand panel L[i, k0:k1]) meet in one L2, and the
    // chunks of the list still go round the XCDs, so its short-to-long ordering loads them evenly (one
    // contiguous eighth per XCD measured -25 % fabric traffic but +6 % time).  Placement is a speed matter
    // only: any bijection is correct.
    const unsigned gx = gridDim.x, nblk = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    // (xcd_map = log2 of the chunk, GPRN_XCD_CHUNK_LOG2 = 4: chunks of 32 / 64 / 128 entries measured 110.4 / 107.5 / 112.7
    // sweeps/s against 112.8 at config 3, 8 entries 113.3 -- DESIGN.md 8; 0 would be grid order)
    const unsigned sh = (unsigned)xcd_map, cmask = (1u << sh) - 1u;
    const unsigned n128 = nblk & ~((8u << sh) - 1u), cx = lin >> 3;
    const unsigned lb = (xcd_map && lin < n128) ? ((((cx >> sh) << 3) + (lin & 7u)) << sh) + (cx & cmask) : lin;
    const unsigned bx = lb % gx, by = lb / gx;
    const TileTask t = tasks[bx / (SM * SN)];
    const int sub = bx % (SM * SN), sr = sub / SN, sc = sub % SN;
    // the slot's four buffer pointers in one scalar load, side by side with the task (no load depends on it)
    double* const* gp = ptrs + (size_t)by * GPRN_NBUF;
    double* const p0 = gp[0]; double* const p1 = gp[1]; double* const p2 = gp[2]; double* const p3 = gp[3];
    auto pick = [&](int b) { return b == 0 ? p0 : (b == 1 ? p1 : (b == 2 ? p2 : p3)); };
    const int c_mode = t.modes & 3;
    const int a_mode = (t.modes >> 2) & 1;
    const int b_mode = (t.modes >> 3) & 1;
    const double* A = pick(t.a_buf) + t.a_off + (a_mode ? (size_t)sr * BM : (size_t)sr * BM * ld);
    const double* B = pick(t.b_buf) + t.b_off + (b_mode ? (size_t)sc * BN : (size_t)sc * BN * ld);
    gptr_t C = (gptr_t)(pick(t.c_buf) + t.c_off) + (size_t)sr * BM * ld + sc * BN;

    // A symmetric update of a DIAGONAL tile (modes bit 4, ensure_tasks): only its lower triangle is ever read (the
    // diagonal-block kernel, the chain's update), so the 64 x 64 quarter above the diagonal is not computed at all and
    // the two quarters on it skip their upper 16 x 16 blocks -- 40 of 64 block products instead of 64 (the exact
    // triangle would be 36).  Round 2 measured 13 % more MFMAs in the bulk launches than the algorithm needs.
    constexpr bool CAN_LOWER = BM == 64 && BN == 64 && TRI == 0 && (TAG == TG_INNER || TAG == TG_NEXT || TAG == TG_BULK || TAG == TG_AHEAD);
    const bool lower = CAN_LOWER && ((t.modes >> 4) & 1);
    // First touch (modes bit 5, the first outer panel's K = 512 update when the caller hands over s = sqrt(d)): the tile
    // of B = I + D^1/2 K D^1/2 is formed from K on the way in instead of being read -- k_build_B then writes only what the
    // first panel's tile steps touch, a quarter of the matrix (run_phase, api_sweep.hip).
    // (64 x 64 workgroups only: the 8-wave 128 x 128 form sits at its 128-register budget and would spill)
    constexpr bool CAN_FT = BM == 64 && BN == 64 && TRI == 0 && (TAG == TG_NEXT || TAG == TG_BULK || TAG == TG_AHEAD);
    const double* ft_K = nullptr;
    const double* ft_sv = nullptr;
    int ft_row = 0, ft_col = 0;
    if (CAN_FT && ft_s && ((t.modes >> 5) & 1)) {
        ft_K = pick(BUF_K) + t.c_off + (size_t)sr * BM * ld + sc * BN;
        ft_sv = ft_s + (size_t)by * ld;
        ft_row = (int)(t.c_off / ld) + sr * BM;
        ft_col = (int)(t.c_off % ld) + sc * BN;
    }
    // ... and in the 8-wave 128 x 128 form (launches of thousands of tasks: T = 128) the same tasks take the instantiation
    // whose diagonal 16 x 16 blocks accumulate from zero (tile_mma SYM: the accuracy of the pivots), all blocks computed
    constexpr bool CAN_SYM128 = BM == 128 && BN == 128 && TRI == 0 && (TAG == TG_INNER || TAG == TG_NEXT);
    if (lower && sr < sc) { /* nothing of this quarter is ever read */ }
    else if (CAN_LOWER && lower && sr == sc)
        tile_mma<BM, BN, WM, WN, TRI, CAN_LOWER, true>(lds, A, B, C, ld, a_mode, b_mode, c_mode, t.klen,
                                                       (sr * BM) >> 4, (sc * BN) >> 4, ft_K, ft_sv, ft_row, ft_col, ft_n);
    else if (CAN_SYM128 && ((t.modes >> 4) & 1))
        tile_mma<BM, BN, WM, WN, TRI, false, false, CAN_SYM128>(lds, A, B, C, ld, a_mode, b_mode, c_mode, t.klen,
                                                                (sr * BM) >> 4, (sc * BN) >> 4);
    else
        tile_mma<BM, BN, WM, WN, TRI, false, true>(lds, A, B, C, ld, a_mode, b_mode, c_mode, t.klen,
                                                   (sr * BM) >> 4, (sc * BN) >> 4, ft_K, ft_sv, ft_row, ft_col, ft_n);
    signal_done(sig_slot, sig_value, then_wait, then_value, wait_timed_out);
}

// Both halves of a tile step's panel in ONE launch: tasks [0, n_l) are L_ik = B_ik X_kk^T (64 x 128 workgroups, B
// triangular), the others X_kc = X_kk R_kc (128 x 64 workgroups, A triangular) -- two workgroups per task in either
// form.  stream3 is the factorisation's second serial chain (synchronise, panel, in-panel update per tile step):
// one launch less per step on it.
// ACC (factor_invert's `prior`: the set-up's factorisation of a prior matrix): the L part by substitution against L_kk
// (trsm_rows16, diag_tile.h: each of the workgroup's four waves 16 of its 64 rows) instead of the product with X_kk.
template <bool ACC>
__global__ __launch_bounds__(256, ACC ? 1 : 2)
void k_tile_panel(const TileTask* __restrict__ tasks, int n_l, double* const* __restrict__ ptrs, int ld,
                  unsigned* sig_slot, unsigned sig_value, const unsigned* then_wait, unsigned then_value,
                  unsigned* wait_timed_out, const unsigned* wait_flag, unsigned wait_value,
                  unsigned* start_flag, unsigned start_value, unsigned* start_flag2)
{
    // (launch_panel: stream3's synchronisation folded into the launch -- the flags of the launches before this one on the
    // stream go up here, and every workgroup waits for the diagonal block itself)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (start_flag) __hip_atomic_store(start_flag, start_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (start_flag2) __hip_atomic_store(start_flag2, start_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    await_flag(wait_flag, wait_value, wait_timed_out);
    __shared__ __attribute__((aligned(16))) double lds[2 * 16 * (64 + 128 + 32)];
    const int ti = blockIdx.x >> 1, sub = blockIdx.x & 1;
    const TileTask t = tasks[ti];
    double* const* gp = ptrs + (size_t)blockIdx.y * GPRN_NBUF;
    double* const p0 = gp[0]; double* const p1 = gp[1]; double* const p2 = gp[2]; double* const p3 = gp[3];
    auto pick = [&](int b) { return b == 0 ? p0 : (b == 1 ? p1 : (b == 2 ? p2 : p3)); };
    const int c_mode = t.modes & 3, a_mode = (t.modes >> 2) & 1, b_mode = (t.modes >> 3) & 1;
    if (ti < n_l) {                                             // rows split
        const double* A = pick(t.a_buf) + t.a_off + (a_mode ? (size_t)sub * 64 : (size_t)sub * 64 * ld);
        const double* B = pick(t.b_buf) + t.b_off;
        gptr_t C = (gptr_t)(pick(t.c_buf) + t.c_off) + (size_t)sub * 64 * ld;
        if constexpr (ACC) {
            // (the task: C = A = tile (i,k) of the B buffer, B = X_kk; L_kk is the same tile of the B buffer)
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            trsm_rows16(lds + wave * TRSM_SCRATCH, C + (size_t)(16 * wave) * ld, (gcptr_t)(pick(t.a_buf) + t.b_off), (gcptr_t)B, ld);
        } else
        tile_mma<64, 128, 2, 2, 1>(lds, A, B, C, ld, a_mode, b_mode, c_mode, t.klen, (sub * 64) >> 4, 0);
    } else {                                                    // columns split
        const double* A = pick(t.a_buf) + t.a_off;
        const double* B = pick(t.b_buf) + t.b_off + (b_mode ? (size_t)sub * 64 : (size_t)sub * 64 * ld);
        gptr_t C = (gptr_t)(pick(t.c_buf) + t.c_off) + sub * 64;
        tile_mma<128, 64, 2, 2, 2>(lds, A, B, C, ld, a_mode, b_mode, c_mode, t.klen, 0, (sub * 64) >> 4);
    }
    signal_done(sig_slot, sig_value, then_wait, then_value, wait_timed_out);
}

int launch_panel(gprn_ctx* c, const TileTask* d_tasks, size_t n_l, size_t n_x, double** d_ptrs, int nbatch, int ld,
                 hipStream_t stream, Signal sig, Await aw, unsigned* raise_at_start, unsigned raise_value, unsigned* raise_at_start2)
{
    if (n_l + n_x == 0 || nbatch == 0) return launch_tiles(c, d_tasks, 0, d_ptrs, nbatch, ld, GPRN_T_PANEL, stream, TS_128x64, sig);
    prof_begin(c, GPRN_T_PANEL, stream);
#define GO_P(ACC) hipLaunchKernelGGL(k_tile_panel<ACC>, dim3((unsigned)(2 * (n_l + n_x)), (unsigned)nbatch), dim3(256), 0, stream, d_tasks, (int)n_l, \
                       (double* const*)d_ptrs, ld, sig.slot, sig.value, sig.then_wait, sig.then_value, \
                       aw.timed_out ? aw.timed_out : sig.timed_out, aw.flag, aw.value, raise_at_start, raise_value, raise_at_start2)
    if (c->acc_now) GO_P(true); else GO_P(false);
#undef GO_P
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ---- the latency chain's two products of a tile step, one 16 x 16 block of the output per WAVE
//   MODE 0:  L_{k+1,k} = B_{k+1,k} X_kk^T, in place over B_{k+1,k}; X_kk lower triangular: block column Q of the
//            result needs k < 16 (Q + 1) only
//   MODE 1:  B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T, lower blocks only (all the diagonal-block kernel reads)
// Operands go (almost) straight from global memory into MFMA operand registers -- lane (fr, fk) takes 16 bytes of row
// fr per load -- and a wave issues at most 32 MFMAs.  The throughput kernel above spends 14 us on the first product (two 64 x 128 workgroups: a 6.8 us MFMA
// stream per wave inside its staging pipeline) and 8 us on the second (four 64 x 64 workgroups); these two are
// latency only: wait, fetch, <= 0.9 us of MFMAs, store, signal.
#ifdef ROWS_STAMPS         // profiles/probes/rows_bench.hip: 100 MHz stamps of workgroup 0, wave 0
__device__ unsigned long long rows_stamps[8];
#define RW_STAMP(i) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) rows_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RW_STAMP(i) do {} while (0)
#endif
// K index of a lane in both kernels: load j (16 bytes per lane) brings k = 8 j + 2 fk and 8 j + 2 fk + 1 of row fr,
// the same split for both operands.
//
// MODE 0, one workgroup of 8 waves per block row P, wave = block column Q.  The row block of B_{k+1,k} (16 x 128,
// operand A of all eight waves, and the memory the result goes to) passes through LDS once -- each wave fetches two
// rows, fully coalesced; eight waves fetching all of it themselves kept the CU's vector memory path busy for 3.5 us
// -- and the X_kk rows come straight from global memory, 16 (Q + 1) columns of them.
#define ROWS_PITCH 136                              // doubles per LDS row: 16-byte reads of 64 lanes spread over all banks
template <bool ARGS>
__global__ __launch_bounds__(512)
void k_chain_l(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int64_t a_off, int64_t b_off,
               unsigned* sig_slot, unsigned sig_value, const unsigned* wait_flag, unsigned wait_value,
               unsigned* wait_timed_out)
{
    __shared__ __attribute__((aligned(16))) double rows[16 * ROWS_PITCH];
    RW_STAMP(0);
    const bool st = pa.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (st) pa.stamps[0] = __builtin_amdgcn_s_memrealtime();
    await_flag(wait_flag, wait_value, wait_timed_out);
    if (st) pa.stamps[1] = __builtin_amdgcn_s_memrealtime();
    RW_STAMP(1);
    const int Q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), P = blockIdx.x;
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    double* const Bm = ARGS ? pa.p[blockIdx.y][0] : ptrs[(size_t)blockIdx.y * GPRN_NBUF + BUF_B];
    double* const Xm = ARGS ? pa.p[blockIdx.y][1] : ptrs[(size_t)blockIdx.y * GPRN_NBUF + BUF_X];
    const int nj = 2 * (Q + 1);
    const double* B = Xm + b_off + (size_t)(16 * Q + fr) * ld + 2 * fk;
    double b[32];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nj) {
            const double2 bv = *(const double2*)(B + 8 * j);
            b[2 * j] = bv.x; b[2 * j + 1] = bv.y;
        }
    {   // rows 2Q, 2Q+1 of the block: lane l takes 16 bytes at column 2 l
        const double* Ar = Bm + a_off + (size_t)(16 * P + 2 * Q) * ld + 2 * lane;
        const double2 r0 = *(const double2*)Ar, r1 = *(const double2*)(Ar + ld);
        *(double2*)(rows + (2 * Q) * ROWS_PITCH + 2 * lane) = r0;
        *(double2*)(rows + (2 * Q + 1) * ROWS_PITCH + 2 * lane) = r1;
    }
    RW_STAMP(2);
    __syncthreads();                 // the block is in LDS: from here on its memory may be overwritten (in place)
    RW_STAMP(3);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* ar = rows + fr * ROWS_PITCH + 2 * fk;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nj) {
            const double2 av = *(const double2*)(ar + 8 * j);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[2 * j], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[2 * j + 1], acc, 0, 0, 0);
        }
    gptr_t C = (gptr_t)(Bm + a_off) + (size_t)(16 * P + fk) * ld + 16 * Q + fr;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) C[(size_t)(4 * tt) * ld] = acc[tt];
    RW_STAMP(4);
#ifdef ROWS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RW_STAMP(5);
#endif
    if (st) pa.stamps[2] = __builtin_amdgcn_s_memrealtime();
    signal_done(sig_slot, sig_value, nullptr, 0, wait_timed_out);
    RW_STAMP(6);
}

// MODE 1, one single-wave workgroup per lower 16 x 16 block (36 per matrix, each on a CU of its own: 32 KiB of
// operands per CU instead of 256)
template <bool ARGS>
__global__ __launch_bounds__(64)
void k_chain_u(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int64_t a_off, int64_t c_off,
               unsigned* sig_slot, unsigned sig_value, const unsigned* wait_flag, unsigned wait_value,
               unsigned* wait_timed_out, unsigned* start_flag, unsigned start_value)
{
    RW_STAMP(0);
    // (launch_tile_rows: the flag of the launch BEFORE this one on the stream, raised here instead of at that one's end --
    // what it wrote is in memory by the time a workgroup of this launch runs)
    if (start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(start_flag, start_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool st = pa.stamps && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0;
    if (st) pa.stamps[0] = __builtin_amdgcn_s_memrealtime();
    await_flag(wait_flag, wait_value, wait_timed_out);
    if (st) pa.stamps[1] = __builtin_amdgcn_s_memrealtime();
    RW_STAMP(1);
    int P = 0;
    while ((P + 1) * (P + 2) / 2 <= (int)blockIdx.x) ++P;
    const int Q = (int)blockIdx.x - P * (P + 1) / 2;
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    double* const Bm = ARGS ? pa.p[blockIdx.y][0] : ptrs[(size_t)blockIdx.y * GPRN_NBUF + BUF_B];
    const double* A = Bm + a_off + (size_t)(16 * P + fr) * ld + 2 * fk;
    const double* B = Bm + a_off + (size_t)(16 * Q + fr) * ld + 2 * fk;
    gptr_t C = (gptr_t)(Bm + c_off) + (size_t)(16 * P + fk) * ld + 16 * Q + fr;
    double a[32], b[32], cin[4];
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double2 av = *(const double2*)(A + 8 * j), bv = *(const double2*)(B + 8 * j);
        a[2 * j] = av.x; a[2 * j + 1] = av.y;
        b[2 * j] = bv.x; b[2 * j + 1] = bv.y;
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) cin[tt] = C[(size_t)(4 * tt) * ld];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every load in flight before the first MFMA
    RW_STAMP(2);
    RW_STAMP(3);
    // the product accumulates from ZERO and is subtracted once: the diagonal blocks hold B's diagonal entries (~1, the
    // update ~d K), and 32 MFMA steps rounding at that magnitude cost the pivots their last digits (tile_mma, SYM)
#pragma unroll
    for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc, 0, 0, 0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) C[(size_t)(4 * tt) * ld] = cin[tt] - acc[tt];
    RW_STAMP(4);
#ifdef ROWS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RW_STAMP(5);
#endif
    if (st) pa.stamps[2] = __builtin_amdgcn_s_memrealtime();
    signal_done(sig_slot, sig_value, nullptr, 0, wait_timed_out);
    RW_STAMP(6);
}

// mode 0 / 1 as above, for tile step k: the operands are tiles (k+1, k), (k, k) [of X] resp. (k+1, k+1), (k+1, k) --
// the first panel and the first update task of the step (ensure_tasks, factor.hip)
// raise_at_start (mode 1): a flag word the launch sets to raise_value as soon as its first workgroup runs -- L_{k+1,k}'s
// flag, when the launch before it on the stream is that product: its own end-of-kernel signal (wait for the stores, barrier,
// release, atomic: 1.7 us between the two launches of every tile step, profiles/r03_chain_stamps_c2.txt) is then not needed
int launch_tile_rows(gprn_ctx* c, int k, double** d_ptrs, int nbatch, int ld, int mode, int fam,
                     hipStream_t stream, Signal sig, Await aw, unsigned* raise_at_start, unsigned raise_value)
{
    if (!stream) stream = c->stream;
    if (nbatch == 0) return GPRN_OK;
    auto toff = [&](int ti, int tj) { return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE; };
    const int64_t a_off = toff(k + 1, k), b_off = mode == 0 ? toff(k, k) : toff(k + 1, k);
    const int64_t c_off = mode == 0 ? toff(k + 1, k) : toff(k + 1, k + 1);
    prof_begin(c, fam, stream);
    double* const* tab = (double* const*)d_ptrs;
    PtrArgs pa;
    pa.stamps = step_stamp_ptr(c, k, mode == 0 ? 1 : 2);
    const bool args = tab_rows(c, d_ptrs, nbatch, &pa);
    unsigned* const tmo = aw.timed_out ? aw.timed_out : sig.timed_out;
#define GO_L(A) hipLaunchKernelGGL((k_chain_l<A>), dim3(GPRN_TILE / 16, (unsigned)nbatch), dim3(512), 0, stream, tab, pa, ld, a_off, \
                                   b_off, sig.slot, sig.value, aw.flag, aw.value, tmo)
#define GO_U(A) hipLaunchKernelGGL((k_chain_u<A>), dim3(36, (unsigned)nbatch), dim3(64), 0, stream, tab, pa, ld, a_off, c_off, \
                                   sig.slot, sig.value, aw.flag, aw.value, tmo, raise_at_start, raise_value)
    if (mode == 0) { if (args) GO_L(true); else GO_L(false); }
    else { if (args) GO_U(true); else GO_U(false); }
#undef GO_L
#undef GO_U
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// LDS one workgroup may ask for on this device (static + dynamic).  A launch beyond it is not refused by the
// runtime: the queue aborts with HSA_STATUS_ERROR_INVALID_ALLOCATION and takes the process down (round 2,
// gpurun_out/r2_b37.err: an LDS-pad experiment on top of the 72 KiB image of the 128 x 128 shape), so every
// launch that carries a pad is checked here first.
size_t lds_limit(int device)
{
    static std::map<int, size_t> known;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    int per_cu = 0, per_block = 0;
    if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess) per_cu = 0;
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) per_block = 0;
    const size_t lim = (size_t)std::max(std::max(per_cu, per_block), 65536);
    known[device] = lim;
    return lim;
}

template <int BM, int BN, int TRI, int TAG>
static bool launch_one(gprn_ctx* c, const TileTask* d_tasks, size_t ntasks, double* const* tab, int nbatch, int ld,
                       size_t dyn, hipStream_t stream, const Signal& sig, const Await& aw)
{
    constexpr size_t static_lds = 2 * 16 * (BM + BN + 32) * sizeof(double);
    if (dyn && static_lds + dyn > lds_limit(c->device)) {
        c->err = "tile launch: " + std::to_string(static_lds) + " B of LDS plus a pad of " + std::to_string(dyn) +
                 " B exceed the " + std::to_string(lds_limit(c->device)) + " B a workgroup may have on this device "
                 "(options bulk_pad_kb / small_pad_kb)";
        return false;
    }
    constexpr int per_task = (GPRN_TILE / BM) * (GPRN_TILE / BN);
    constexpr int NW = (BM == 128 && BN == 128) ? 8 : 4;       // the throughput shape runs on 8 waves
    hipLaunchKernelGGL((k_tile_gemm<BM, BN, NW, TRI, TAG>), dim3((unsigned)ntasks * per_task, (unsigned)nbatch),
                       dim3(64 * NW), dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait,
                       sig.then_value, aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out, GPRN_XCD_CHUNK_LOG2,
                       c->ft_s_now, c->N, c->start_flag_now, c->start_value_now);
    return true;
}

int launch_tiles(gprn_ctx* c, const TileTask* d_tasks, size_t ntasks, double** d_ptrs,
                 int nbatch, int ld, int fam, hipStream_t stream, int shape, Signal sig, Await aw, int tag)
{
    if (!stream) stream = c->stream;
    if (ntasks == 0 || nbatch == 0) {
        // No kernel: keep what the launch would have done to the stream's order.  The waits it would have
        // made become stream waits, and the flag (raised only by signals that carry a value: 0 means "count
        // the workgroups", and flags never go down) is written from the stream.  Signals and waits exist in
        // the flag schedule only, which requires stream memory operations (factor_use_flags).
        if (aw.flag) HIP_TRY(c, hipStreamWaitValue32(stream, (void*)aw.flag, aw.value, hipStreamWaitValueGte, 0xffffffffu));
        if (sig.slot && sig.value) HIP_TRY(c, hipStreamWriteValue32(stream, sig.slot + 1, sig.value, 0));
        if (sig.slot && sig.then_wait)
            HIP_TRY(c, hipStreamWaitValue32(stream, (void*)sig.then_wait, sig.then_value, hipStreamWaitValueGte, 0xffffffffu));
        return GPRN_OK;
    }
    // (ACC: the L part of a panel never runs as a product -- k_tile_panel<true>, same grid, same flags)
    if (c->acc_now && shape == TS_64x128_BTRI && tag == TG_PANEL)
        return launch_panel(c, d_tasks, ntasks, 0, d_ptrs, nbatch, ld, stream, sig, aw, c->start_flag_now, c->start_value_now);
    prof_begin(c, fam, stream);
    // Bulk launches on the look-ahead stream (the K = 512 trailing updates, the X^T X product) ask for 16 KiB of unused
    // dynamic LDS on top of their image: two 64 x 64 workgroups per CU instead of three (one 128 x 128 instead of two).
    // Workgroups are never preempted and stream priorities do not reorder dispatch, so this is what keeps part of every
    // CU's LDS and wave slots open for the latency chain's kernels (measured +4 % sweeps/s at config 3; a CU mask for the
    // bulk stream measured worse; pads of 0 / 8 / 24 KiB 112.0 / 113.5 / 113.7 against 114.2 with 16).  Launches over one
    // or two matrices (node half-sweep, sharded runs) ask for 64 KiB: one such workgroup per CU -- those phases are bound by
    // their chains, not by the bulk.  Options "bulk_pad_kb" / "small_pad_kb" override (tests).
    const int kb = nbatch <= 2 ? (c->pad_small_kb_opt >= 0 ? c->pad_small_kb_opt : 64) : (c->pad_kb_opt >= 0 ? c->pad_kb_opt : 16);
    const bool padded_fam = fam == GPRN_T_UPDATE || fam == GPRN_T_UPDATE_AHEAD || fam == GPRN_T_LAUUM;
    const size_t dyn = (stream == c->stream2 && padded_fam) ? (size_t)kb * 1024 : 0;
    double* const* tab = (double* const*)d_ptrs;
#define GO(BM, BN, TRI, TAG) fits = launch_one<BM, BN, TRI, TAG>(c, d_tasks, ntasks, tab, nbatch, ld, dyn, stream, sig, aw)
    bool known = true, fits = true;
    switch (shape * 8 + tag) {
    // panel products (K = 128 against the triangular X_kk)
    case TS_64x128_BTRI * 8 + TG_PANEL: GO(64, 128, 1, TG_PANEL); break;
    case TS_128x64_ATRI * 8 + TG_PANEL: GO(128, 64, 2, TG_PANEL); break;
    // in-panel updates, K = 128
    case TS_64x64 * 8 + TG_INNER: GO(64, 64, 0, TG_INNER); break;
    case TS_128x128 * 8 + TG_INNER: GO(128, 128, 0, TG_INNER); break;
    // next-panel part of an outer update, K = 512
    case TS_64x64 * 8 + TG_NEXT: GO(64, 64, 0, TG_NEXT); break;
    case TS_128x128 * 8 + TG_NEXT: GO(128, 128, 0, TG_NEXT); break;
    // bulk of an outer update, K = 512
    case TS_64x64 * 8 + TG_BULK: GO(64, 64, 0, TG_BULK); break;
    case TS_128x128 * 8 + TG_BULK: GO(128, 128, 0, TG_BULK); break;
    // ... its look-ahead part (what the next panel's outer update writes again), a launch of its own
    case TS_64x64 * 8 + TG_AHEAD: GO(64, 64, 0, TG_AHEAD); break;
    case TS_128x128 * 8 + TG_AHEAD: GO(128, 128, 0, TG_AHEAD); break;
    // X^T X, prediction products, diagnostics
    case TS_128x128 * 8 + TG_MISC: GO(128, 128, 0, TG_MISC); break;
    case TS_64x64 * 8 + TG_MISC: GO(64, 64, 0, TG_MISC); break;
    case TS_64x128 * 8 + TG_MISC: GO(64, 128, 0, TG_MISC); break;
    case TS_128x64 * 8 + TG_MISC: GO(128, 64, 0, TG_MISC); break;
    default: known = false;
    }
#undef GO
    prof_end(c);
    if (!known) { c->err = "launch_tiles: no kernel for this shape/tag"; return GPRN_E_ARG; }
    if (!fits) return GPRN_E_ARG;
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ---- diagnostic: the fp64 MFMA issue ceiling of this device ------------------------------
// Every wave issues `iters` x 16 independent v_mfma_f64_16x16x4_f64 back to back from
// registers (no memory traffic); 4 waves per workgroup, `wg_per_cu` workgroups per CU.
__global__ __launch_bounds__(256, 2)
void k_mfma_peak(double* __restrict__ out, int iters)
{
    v4d acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-12;
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int gprn_test_mfma_peak(gprn_ctx* c, int wg_per_cu, int iters, double* tflops)
{
    if (!c || wg_per_cu < 1 || iters < 1 || !tflops) return GPRN_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    const int nwg = prop.multiProcessorCount * wg_per_cu;
    double* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)nwg * 256 * sizeof(double)));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma_peak, dim3(nwg), dim3(256), 0, c->stream, d, iters);   // warm-up
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(k_mfma_peak, dim3(nwg), dim3(256), 0, c->stream, d, iters);
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    *tflops = (double)nwg * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
    return GPRN_OK;
}
