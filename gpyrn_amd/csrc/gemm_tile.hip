// Batched 128x128 tile contraction on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Every O(N^3) flop of the sweep goes through this one kernel: the trailing
// SYRK of the blocked Cholesky, the row updates that build L^-1 alongside it,
// the panel solves (as products with the inverted diagonal block) and the
// X^T X product behind quirk Q1 (see factor.hip for the task lists).
//
// Work decomposition: one 256-thread workgroup (4 waves, 2x2) per 128x128
// output tile, each wave a 64x64 sub-tile = 4x4 MFMA tiles of 16x16, i.e. 16
// independent accumulators (128 VGPRs) -- enough to issue the 64-cycle f64 MFMA
// back to back.  K is streamed in chunks of 16 through a double-buffered LDS
// image (global -> registers -> LDS, one barrier per chunk); global loads of
// chunk c+1 are in flight while chunk c is multiplied.
//
// LDS images (conflict-free for the one-f64-per-lane MFMA operand fetch,
// ds_read_b64, banks = (addr/4)%64):
//   operand stored k-contiguous in memory  -> image [row][k],  pitch 18 doubles
//   operand stored row-contiguous in memory-> image [k][row],  pitch 144 doubles
// both 2304 doubles per operand per stage; 4 x 18 KiB = 72 KiB per workgroup,
// two workgroups per CU.
#include "gprn_internal.h"

#include <stdlib.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const GPRN_GLOBAL v2d* gv2d_t;

// Output tile of one workgroup: BM x BN in {64,128}^2.  A 128x128 task is cut into
// (128/BM) x (128/BN) workgroups (sub-tile index = blockIdx.x % that).  128x128 is the
// throughput shape (bulk updates); the smaller shapes put a latency-bound launch -- a few
// dozen tasks with K = 128 on the factorisation's critical path -- on 2-4x as many CUs:
//   64x128 (rows split)  is safe for in-place panel tasks whose C tile is their A operand,
//   128x64 (cols split)  for in-place tasks whose C tile is their B operand,
//   64x64                for everything that is not in place.
// TRI: an operand of every task of the launch is the inverted diagonal block X_kk (lower triangular,
// explicit zeros above): 1 = it is B, transposed (panel product L_ik = B_ik X_kk^T: k <= n),
// 2 = it is A (X_kc = X_kk R_kc: k <= m).  K-chunks that only meet the zero half of a 16-wide
// block are skipped -- 28 of 64 block products.
template <int BM, int BN, int TRI = 0>
__global__ __launch_bounds__(256, 2)
void k_tile_gemm(const TileTask* __restrict__ tasks, double* const* __restrict__ ptrs, int ld,
                 unsigned* sig_slot, unsigned sig_value, const unsigned* then_wait, unsigned then_value,
                 const unsigned* wait_flag, unsigned wait_value, unsigned* wait_timed_out)
{
    await_flag(wait_flag, wait_value, wait_timed_out);
    constexpr int SM = GPRN_TILE / BM, SN = GPRN_TILE / BN;     // sub-tiles per task
    constexpr int MI = BM / 32, NI = BN / 32;                   // 16x16 MFMA tiles per wave (2x2 waves)
    constexpr int A_DOUBLES = 16 * BM + 256, B_DOUBLES = 16 * BN + 256;   // >= BM*18 and 16*(BM+16)
    constexpr int A_IT = BM / 32, B_IT = BN / 32;               // 16-byte loads per thread per chunk
    __shared__ __attribute__((aligned(16))) double lds[2 * (A_DOUBLES + B_DOUBLES)];

    const TileTask t = tasks[blockIdx.x / (SM * SN)];
    const int sub = blockIdx.x % (SM * SN), sr = sub / SN, sc = sub % SN;
    double* const* gp = ptrs + (size_t)blockIdx.y * GPRN_NBUF;
    const int c_mode = t.modes & 3;
    const int a_mode = (t.modes >> 2) & 1;
    const int b_mode = (t.modes >> 3) & 1;
    gcptr_t A = (gcptr_t)(gp[t.a_buf] + t.a_off) + (a_mode ? (size_t)sr * BM : (size_t)sr * BM * ld);
    gcptr_t B = (gcptr_t)(gp[t.b_buf] + t.b_off) + (b_mode ? (size_t)sc * BN : (size_t)sc * BN * ld);
    gptr_t C = (gptr_t)(gp[t.c_buf] + t.c_off) + (size_t)sr * BM * ld + sc * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- global -> LDS staging geometry (pairs of doubles along the contiguous dim)
    // k-contiguous operand: image [row][k], pitch 18;  row-contiguous: image [k][row], pitch rows+16
    const int a_shift = a_mode ? (BM == 128 ? 6 : 5) : 3, b_shift = b_mode ? (BN == 128 ? 6 : 5) : 3;
    const int a_pitch = a_mode ? BM + 16 : 18, b_pitch = b_mode ? BN + 16 : 18;
    const size_t a_step = a_mode ? (size_t)GPRN_KC * ld : (size_t)GPRN_KC;
    const size_t b_step = b_mode ? (size_t)GPRN_KC * ld : (size_t)GPRN_KC;
    size_t a_g[A_IT], b_g[B_IT];
    int a_l[A_IT], b_l[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int pi = tid + 256 * it;
        const int as = pi >> a_shift, af = pi & ((1 << a_shift) - 1);
        a_g[it] = (size_t)as * ld + 2 * af;
        a_l[it] = as * a_pitch + 2 * af;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int pi = tid + 256 * it;
        const int bs = pi >> b_shift, bf = pi & ((1 << b_shift) - 1);
        b_g[it] = (size_t)bs * ld + 2 * bf;
        b_l[it] = bs * b_pitch + 2 * bf;
    }
    // ---- MFMA operand fetch geometry: lane holds A[row = l&15][k = l>>4], B[k = l>>4][col = l&15]
    const int fr = lane & 15, fk = lane >> 4;
    const int a_rs = a_mode ? 1 : 18, a_ks = a_mode ? BM + 16 : 1;
    const int b_rs = b_mode ? 1 : 18, b_ks = b_mode ? BN + 16 : 1;
    const int mb16 = (sr * BM + wr * (BM / 2)) >> 4, nb16 = (sc * BN + wc * (BN / 2)) >> 4;   // wave's first 16-blocks
    const int a_frag = (wr * (BM / 2) + fr) * a_rs + fk * a_ks;
    const int b_frag = (wc * (BN / 2) + fr) * b_rs + fk * b_ks;

    // C -= A.B runs as D = (-A).B + C with the accumulators preloaded from C: the tile is read
    // once, up front and all loads back to back (a read-modify-write epilogue serialises into
    // dependent load->store round trips), and the epilogue is stores only.
    const double a_sign = (c_mode == CM_SET) ? 1.0 : -1.0;
    gptr_t Cw = C + (size_t)(wr * (BM / 2) + fk) * ld + wc * (BN / 2) + fr;
    v4d acc[MI][NI];

    const int nchunks = t.klen / GPRN_KC;
    // (de-phasing co-resident workgroups by half a chunk and s_setprio around the MFMA clusters
    // measured no gain; PMC: MFMA pipe busy 78 %
    // of the cycles at an effective 2.13 GHz, no LDS bank conflicts -- DESIGN.md section 8)
    // Software pipeline over K-chunks, one barrier per chunk:
    //   registers hold chunk c+1 (requested during chunk c-1 ... c), LDS stage c&1 holds chunk c.
    //   While chunk c is multiplied, chunk c+1 is written to the other LDS stage after the second
    //   of the four k4-steps (its global loads have had >2000 cycles), and chunk c+2 is requested.
    //   The barrier at the end of the chunk publishes stage (c+1)&1 and retires the reads of c&1.
    // The LDS staging therefore overlaps the MFMA stream instead of sitting between two chunks.
    v2d ra[A_IT], rb[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) ra[it] = *(gv2d_t)(A + a_g[it]);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(gv2d_t)(B + b_g[it]);
    // The C tile is requested AFTER the first operand chunk: memory returns in order, so the
    // LDS staging below waits only for the chunk, and the first MFMA of each accumulator only for
    // its own four values -- most of the 128 KiB tile streams in behind the first MFMAs.
    __builtin_amdgcn_sched_barrier(0);
    if (c_mode == CM_SUB) {                      // one uniform branch around all 16*MI*NI/4 loads
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = Cw[(size_t)(i * 16 + 4 * r) * ld + j * 16];
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        double* sA = lds;
        double* sB = sA + A_DOUBLES;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) *reinterpret_cast<v2d*>(sA + a_l[it]) = ra[it] * a_sign;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) *reinterpret_cast<v2d*>(sB + b_l[it]) = rb[it];
    }
    if (nchunks > 1) {
        A += a_step;
        B += b_step;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) ra[it] = *(gv2d_t)(A + a_g[it]);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[it] = *(gv2d_t)(B + b_g[it]);
    }
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const double* sA = lds + (c & 1) * (A_DOUBLES + B_DOUBLES);
        const double* sB = sA + A_DOUBLES;
        double* nA = lds + ((c + 1) & 1) * (A_DOUBLES + B_DOUBLES);
        double* nB = nA + A_DOUBLES;
        // Operand fragments of k4-step ks+1 are fetched from LDS before the MFMAs of step ks are
        // issued: a lone wave per SIMD (bulk launches run at one workgroup per CU) then does not
        // drain the matrix pipe at every step waiting for its own ds_reads (44 -> 46 TF at K = 512
        // and one workgroup per CU).  Carrying the fetch-ahead across the chunk boundary as well
        // (barrier after the third step, first fragments of the next chunk read during the fourth)
        // measured no further gain.
        double af[2][MI], bf[2][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = sA[a_frag + i * 16 * a_rs];
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[0][j] = sB[b_frag + j * 16 * b_rs];
#pragma unroll
        for (int ks = 0; ks < GPRN_KC / 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < GPRN_KC / 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[nxt][i] = sA[a_frag + i * 16 * a_rs + (ks + 1) * 4 * a_ks];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[nxt][j] = sB[b_frag + j * 16 * b_rs + (ks + 1) * 4 * b_ks];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (TRI == 1 && c > nb16 + j) continue;     // wave-uniform
                    if (TRI == 2 && c > mb16 + i) continue;
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                }
            if (ks == 1 && c + 1 < nchunks) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < A_IT; ++it) *reinterpret_cast<v2d*>(nA + a_l[it]) = ra[it] * a_sign;
#pragma unroll
                for (int it = 0; it < B_IT; ++it) *reinterpret_cast<v2d*>(nB + b_l[it]) = rb[it];
                if (c + 2 < nchunks) {
                    A += a_step;
                    B += b_step;
#pragma unroll
                    for (int it = 0; it < A_IT; ++it) ra[it] = *(gv2d_t)(A + a_g[it]);
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) rb[it] = *(gv2d_t)(B + b_g[it]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cw[(size_t)(i * 16 + 4 * r) * ld + j * 16] = acc[i][j][r];
    signal_done(sig_slot, sig_value, then_wait, then_value, wait_timed_out);
}

int launch_tiles(gprn_ctx* c, const TileTask* d_tasks, size_t ntasks, double** d_ptrs,
                 int nbatch, int ld, int fam, hipStream_t stream, int shape, Signal sig, Await aw)
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
    prof_begin(c, fam, stream);
    // Bulk launches on the look-ahead stream ask for 16 KiB of unused dynamic LDS on top of the
    // 72 KiB image: one workgroup per CU instead of two.  Workgroups are never preempted and stream
    // priorities do not reorder dispatch, so this is what keeps half of every CU's LDS and wave
    // slots open for the latency chain's kernels (measured +4 % sweeps/s at config 3; a CU mask
    // for the bulk stream measured worse).  GPRN_BULK_PAD_KB overrides.
    static int pad_kb = -1;
    if (pad_kb < 0) { const char* e = getenv("GPRN_BULK_PAD_KB"); pad_kb = e ? atoi(e) : 16; }
    static int pad_fams = -1;                      // GPRN_PAD_FAMS: bit per family that gets the pad (default: the
                                                   // bulk update and the X^T X product; the next-panel launches measured
                                                   // slightly better without it)
    if (pad_fams < 0) { const char* e = getenv("GPRN_PAD_FAMS"); pad_fams = e ? atoi(e) : ((1 << GPRN_T_UPDATE) | (1 << GPRN_T_LAUUM)); }
    static int pad_all = -1;                       // GPRN_PAD_ALL=1 (probes): pad on every stream
    if (pad_all < 0) { const char* e = getenv("GPRN_PAD_ALL"); pad_all = e ? atoi(e) : 0; }
    // Launches over one or two matrices (node half-sweep, sharded runs) ask for 64 KiB instead: with
    // 104 KiB taken, the diagonal-block kernel (67 KiB) cannot land on a CU that runs a bulk workgroup at
    // all and runs undisturbed on the next CU that comes free -- those phases are bound by the chain, not
    // by the bulk (+3 % sweeps/s at config 3).  GPRN_PAD_SMALL_KB / GPRN_PAD_SMALL_BATCH override.
    static int pad_small_kb = -1, pad_small_batch = -1;
    if (pad_small_kb < 0) { const char* e = getenv("GPRN_PAD_SMALL_KB"); pad_small_kb = e ? atoi(e) : 64; }
    if (pad_small_batch < 0) { const char* e = getenv("GPRN_PAD_SMALL_BATCH"); pad_small_batch = e ? atoi(e) : 2; }
    const int kb = nbatch <= pad_small_batch ? pad_small_kb : pad_kb;
    const size_t dyn = ((stream == c->stream2 || pad_all) && ((pad_fams >> fam) & 1)) ? (size_t)kb * 1024 : 0;
    double* const* tab = (double* const*)d_ptrs;
    switch (shape) {
    case TS_64x64:
        hipLaunchKernelGGL((k_tile_gemm<64, 64>), dim3((unsigned)ntasks * 4, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
        break;
    case TS_64x128:
        hipLaunchKernelGGL((k_tile_gemm<64, 128>), dim3((unsigned)ntasks * 2, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
        break;
    case TS_128x64:
        hipLaunchKernelGGL((k_tile_gemm<128, 64>), dim3((unsigned)ntasks * 2, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
        break;
    case TS_64x128_BTRI:
        hipLaunchKernelGGL((k_tile_gemm<64, 128, 1>), dim3((unsigned)ntasks * 2, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
        break;
    case TS_128x64_ATRI:
        hipLaunchKernelGGL((k_tile_gemm<128, 64, 2>), dim3((unsigned)ntasks * 2, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
        break;
    default:
        hipLaunchKernelGGL((k_tile_gemm<128, 128>), dim3((unsigned)ntasks, (unsigned)nbatch), dim3(256),
                           dyn, stream, d_tasks, tab, ld, sig.slot, sig.value, sig.then_wait, sig.then_value,
                           aw.flag, aw.value, aw.timed_out ? aw.timed_out : sig.timed_out);
    }
    prof_end(c);
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
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    return GPRN_OK;
}
