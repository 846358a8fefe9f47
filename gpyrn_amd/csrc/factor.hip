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
//   diag   : L_kk = chol(B_kk), X_kk = L_kk^-1            one workgroup, LDS (diag_tile.h)
//   panel  : L_ik = B_ik X_kk^T (i>k);  X_kc = X_kk R_kc (c<k)   tile GEMMs, K=128
//   update : B_ij -= L_ik L_jk^T (i>=j>k);  R_ic -= L_ik X_kc (i>k, c<=k)
// where R (the running right-hand side of L X = I) lives in X's own tiles and
// tile (i,c) is first written, not accumulated, at step k == c.  Per step the
// update touches (T-1-k)(T-k)/2 + (T-1-k)(k+1) tiles -- roughly constant until
// the tail, unlike POTRF alone.  Flops: N^3/3 + N^3/3.
//
// ONE schedule (factor_invert_launches), on device-side flags or -- the same launch sequence -- on HIP events.  The
// schedules that were tried beside it and measured slower (a persistent dependency-queue worker kernel, a block
// schedule with transposed mirrors, persistent chain workgroups, left-looking and column-by-column forms of the updates)
// are recorded with their numbers in DESIGN.md 5b / 5c / 8 and live in the history of this file (round 3).
#include "gprn_internal.h"
#include "tile_mma.h"
#include "diag_tile.h"
#include "vecops.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>

// PTRS: the two pointers per matrix come as kernel arguments (launch_diag), else from the table
// (one wave per SIMD: the register-resident tile needs ~290 VGPRs per lane; without the second bound the compiler sizes
// the allocation for the three workgroups per CU the LDS would allow and spills)
template <bool ARGS, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_diag_block(double* const* __restrict__ ptrs, PtrArgs pa, int ld, int kblk, int* __restrict__ info,
                  unsigned* sig_slot, unsigned sig_value, const unsigned* wait_flag, unsigned wait_value,
                  unsigned* wait_timed_out, int nph)
{
    __shared__ __attribute__((aligned(16))) double lds[ACC ? DIAG_LDS_DOUBLES_ACC : DIAG_LDS_DOUBLES];
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[0] = __builtin_amdgcn_s_memrealtime();
    await_flag(wait_flag, wait_value, wait_timed_out);
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[1] = __builtin_amdgcn_s_memrealtime();
    const int slot = blockIdx.x;
    const size_t off = ((size_t)kblk * GPRN_TILE) * ld + (size_t)kblk * GPRN_TILE;
    double* const Bm = ARGS ? pa.p[slot][0] : ptrs[(size_t)slot * GPRN_NBUF + BUF_B];
    double* const Xm = ARGS ? pa.p[slot][1] : ptrs[(size_t)slot * GPRN_NBUF + BUF_X];
    // (nph: the 16-column phases of this tile that hold data -- the last tile of a matrix whose size is not a multiple of 128
    // ends in identity padding, whose factor diag_tile writes without running the phases)
    diag_tile<ACC>(lds, (gptr_t)(Bm + off), (gptr_t)(Xm + off), ld, info, slot, kblk * GPRN_TILE, nph);
    if (pa.stamps && blockIdx.x == 0 && threadIdx.x == 0) pa.stamps[2] = __builtin_amdgcn_s_memrealtime();
    signal_done(sig_slot, sig_value, nullptr, 0, nullptr);
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
    if (nbatch > GPRN_ARG_SLOTS) return false;
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
    // Little work in total (batch x tiles <= GPRN_LAT_MAX, the latency set's problems): most CUs are idle, and with all
    // the LDS a workgroup may have (113 KiB of unused dynamic LDS on top of its 46.6) the workgroup only lands on a CU that
    // runs nothing else which uses LDS -- no co-resident MFMA waves of the side stream's tile kernels on its SIMDs.
    // Config 2 (N = 2048, one matrix per phase): 686 -> 741 sweeps/s.  On a loaded device it waits for such a CU as long
    // as the neighbours would have cost (config 3 with the pad in the node phase: 109.8 vs 110.1), hence the limit.
    size_t dyn = 0;
    if (nbatch * c->T <= GPRN_LAT_MAX)
        dyn = std::min<size_t>((size_t)113 * 1024, lds_limit(c->device) - (c->acc_now ? DIAG_LDS_DOUBLES_ACC : DIAG_LDS_DOUBLES) * sizeof(double));
    pa.stamps = step_stamp_ptr(c, kblk, 0);
    // rows of data in this tile: all 128 but in the last tile of a ragged matrix (ld is the context's: the diagnostic entry
    // points factor whole tiles)
    const int rows_here = ld == c->ld ? std::min(GPRN_TILE, c->N - kblk * GPRN_TILE) : GPRN_TILE;
    const int nph = std::max(1, (rows_here + 15) / 16);
#define GO_D(ARGS, ACC) hipLaunchKernelGGL((k_diag_block<ARGS, ACC>), dim3(nbatch), dim3(256), dyn, stream, (double* const*)d_ptrs, pa, ld, kblk, \
                           d_info, sig.slot, sig.value, aw.flag, aw.value, aw.timed_out, nph)
    if (tab_rows(c, d_ptrs, nbatch, &pa)) { if (c->acc_now) GO_D(true, true); else GO_D(true, false); }
    else { if (c->acc_now) GO_D(false, true); else GO_D(false, false); }
#undef GO_D
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
// The outer update is split into the part the next panel needs ("first", "next": its
// columns of B, its rows of R) and the rest, so the next panel's latency chain
// can run while the rest streams on a second HIP stream.
// Two sets of lists: [0] outer panels of GPRN_OUTER tiles (throughput), [1] of GPRN_OUTER_SMALL for problems with little
// work in total (batch x tiles <= GPRN_LAT_MAX: fewer joins of the bulk stream on the chain).
int ensure_tasks(gprn_ctx* c)
{
    const int T = c->T, ld = c->ld;
    if (c->tasks_T == T && c->d_tasks) return GPRN_OK;
    std::vector<TileTask>& v = c->h_tasks;
    v.clear();
    for (int set = 0; set < 2; ++set) {
    const int outer = set ? GPRN_OUTER_SMALL : GPRN_OUTER;
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
            // first `ncol1` tasks are what has to be done before the chain may go on (factor_invert_launches launches
            // them on their own where the others must wait for an outer update).
            // (diagonal tiles of the symmetric updates: lower blocks only, tile_modes' last argument)
            auto is_crit = [&](int i, int j) { return (i == k + 1 && j == k + 1) || (i == k + 2 && (j == k + 1 || j == k + 2)); };
            for (int pass = 0; pass < 2; ++pass) {
                for (int j = k + 1; j < std::min(T, k1 + outer); ++j)
                    for (int i = j; i < (j < k1 ? T : std::min(T, j + 2)); ++i) {
                        if (is_crit(i, j) != (pass == 0)) continue;
                        v.push_back(TileTask{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                                             BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0, i == j)});
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
        gprn_ctx::OuterRange o{k0, k1, 0, 0, 0, 0, 0, 0, 0};
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
                                         (uint8_t)(tile_modes(CM_SUB, 0, 0, i == j) | (k0 == 0 ? 32 : 0))});
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
            if (pass == 0) { o.first0 = begin; o.nfirst = v.size() - begin; }
            else if (pass == 1) { o.next0 = begin; o.nnext = v.size() - begin; }
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

#define GPRN_FLAG_KINDS 10          // flag kinds per tile step / outer panel (factor_invert_launches)

// Flags or events for this context?  Kernels that wait for other kernels need those to be able to run
// beside them: every switch that serialises kernels or starves the hardware queues means events.
//   rocprofv3 --pmc (ROCPROF_COUNTER_COLLECTION=1), AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING,
//   GPU_MAX_HW_QUEUES < 4 (three streams of this context + the null stream), no stream memory operations.
// GPRN_FLAGS=0/1 overrides; gprn_set_option(ctx, "flags", v) sets it per context; a time-out latches 0.
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


// The launch schedule.  Three serial sequences and a background, on four streams:
//   chain   (s0): per tile step k   diag(k)  ->  L_{k+1,k}  ->  B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T
//                 -- every tile step looks the same to it, panel boundaries included: the two tiles the next panel starts
//                 with are kept up to date step by step (ensure_tasks), so no K = 512 update sits on the chain;
//   side    (s1): per tile step the other panel tiles (one launch, k_tile_panel), then the other in-panel updates (K = 128);
//                 per outer panel the "first" part of its K = 512 update;
//   next    (s4): per outer panel the "next" part (T <= 64; on the bulk stream beyond: a "rest" launch there runs 11 ms);
//   bulk    (s2): per outer panel "rest" in two launches (look-ahead part, then the others), and the phase's row
//                 reductions over X as its rows become final (run_phase's rows_final hook).
// Order of read-modify-writes on a tile is kept by flags (use_flags: 32-bit words in device memory -- the chain's small
// kernels raise and poll them in-kernel, larger launches are bracketed by one-thread kernels or stream memory operations)
// or, the same launch sequence, by HIP events (serialising tools, no stream memory operations, or after a time-out).
// Flags only grow: a call waits for its own epoch.
static int factor_invert_launches(gprn_ctx* c, int nbatch, int set)
{
    int rc;
    hipStream_t s0 = c->stream, s1 = c->stream3, s2 = c->stream2;
    auto shape_upd = [&](size_t n) { return n * (size_t)nbatch > GPRN_FEW_TASKS ? TS_128x128 : TS_64x64; };
    auto tiles = [&](size_t first, size_t n, hipStream_t st, int shape, int fam = GPRN_T_PANEL,
                     Signal sig = Signal{nullptr, 0, nullptr, 0, nullptr}, Await aw = Await{nullptr, 0, nullptr}, int tag = TG_INNER) {
        return launch_tiles(c, c->d_tasks + first, n, c->d_ptrs, nbatch, c->ld, fam, st, shape, sig, aw, tag);
    };
    const int use_flags = factor_use_flags(c);
    auto side_stamp = [&](int k, int i) {          // GPRN_STEP_STAMPS=2: the clock on stream3 at this point of step k
        if (c->side_stamps) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s1, c->side_stamps + (size_t)k * 8 + i);
    };
    enum { F_DIAG = 0, F_MINIL, F_INNER, F_PANEL, F_NEXT, F_REST, F_FIRST, F_XW, F_RESTA, F_TAIL, F_KINDS };
    static_assert(F_KINDS == GPRN_FLAG_KINDS, "factor_check_waits decodes the flag table by GPRN_FLAG_KINDS");
    if (use_flags && c->sig_T < c->T) {
        if (c->d_sig) hipFree(c->d_sig);
        c->d_sig = nullptr;
        // [T][F_KINDS] pairs {counter, flag}, then: sticky time-out word, budget of one wait, which flag timed out
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
    hipEvent_t events[F_KINDS] = {c->ev_diag, c->ev_minil, c->ev_inner, c->ev_panel, c->ev_next, c->ev_rest, c->ev_first,
                                  nullptr, c->ev_resta, c->ev_tail};
    auto slot = [&](int idx, int kind) { return c->d_sig + ((size_t)idx * F_KINDS + kind) * 2; };
    unsigned* const timed_out = c->d_sig ? c->d_sig + (size_t)c->sig_T * F_KINDS * 2 : nullptr;
    const Await noaw{nullptr, 0, nullptr};
    const Signal nosig{nullptr, 0, nullptr, 0, nullptr};
    auto in_kernel = [&](int idx, int kind) {      // the launch raises the flag itself when its last workgroup retires
        return use_flags ? Signal{slot(idx, kind), epoch, nullptr, 0, nullptr} : nosig;
    };
    auto in_kernel_wait = [&](int idx, int kind) { return Await{slot(idx, kind) + 1, epoch, timed_out}; };
    int inner_raises = 0;
    auto withheld = [&](int kind) {                // test hook (gprn_set_option "withhold_inner")
        return use_flags && kind == F_INNER && c->withhold_inner > 0 && ++inner_raises == c->withhold_inner;
    };
    auto raise = [&](hipStream_t st, int idx, int kind) {
        if (withheld(kind)) return hipSuccess;
        return use_flags ? hipStreamWriteValue32(st, slot(idx, kind) + 1, epoch, 0) : hipEventRecord(events[kind], st);
    };
    auto await = [&](hipStream_t st, int idx, int kind) {
        return use_flags ? hipStreamWaitValue32(st, slot(idx, kind) + 1, epoch, hipStreamWaitValueGte, 0xffffffffu)
                         : hipStreamWaitEvent(st, events[kind], 0);
    };
    // one wave on a stream: raise a flag (or none), then wait for one (or none), bounded by the budget -- a stream write
    // and a stream wait of the runtime are 4-5 us kernels each, and the latter has no time-out
    auto flag_sync = [&](hipStream_t st, unsigned* up, const unsigned* wait_for) -> int {
        hipLaunchKernelGGL(k_flag_sync, dim3(1), dim3(64), 0, st, up, epoch, wait_for, epoch, timed_out);
        HIP_TRY(c, hipGetLastError());
        return GPRN_OK;
    };
    auto flag_multi = [&](hipStream_t st, const FlagOps& ops) -> int {
        hipLaunchKernelGGL(k_flag_multi, dim3(1), dim3(64), 0, st, ops, epoch, timed_out);
        HIP_TRY(c, hipGetLastError());
        return GPRN_OK;
    };
    // A launch whose EVERY workgroup polls at its head must not be able to fill the device: workgroups are never
    // preempted, so once pollers hold every slot a producer that is not resident yet never becomes so and the flag
    // never rises (round 3, config 5's shape: the last tile step's 2 (T - 1) x 15 = 3810 workgroups polled for an
    // update of stream3 that was still queued behind its panel launch -- DESIGN.md 9).  Up to half a workgroup per CU
    // they leave room on every CU whatever else they are; above that a one-wave kernel waits instead.
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || n_cu <= 0) n_cu = 64;
    auto may_poll = [&](size_t nwg) { return use_flags && nwg * 2 <= (size_t)n_cu; };

    int rest_J = -1, next_J = -1, first_J = -1;    // outer panels whose rest / next / first update is not joined yet
    int inner_k = -1;                              // tile step whose F_INNER flag stream3 still has to raise
    int pending_outer = -1;                        // outer panel whose trailing update is not enqueued yet
    unsigned* pending_up = nullptr;                // a flag the next folded panel launch on stream3 raises at its start
    bool tail_on_s2 = false;                       // rows_final ran on the bulk stream: joined at the end
    // "rest" as two launches with "next" on a stream of its own, beside the previous panel's whole "rest": +2.1 % sweeps/s
    // at N = 4096 and 8192; at N = 16384, where a "rest" launch runs for 11 ms, -0.7 %: up to 64 tile steps
    const bool sr = c->stream4 && c->T <= 64;
    hipStream_t sn = sr ? c->stream4 : s2;
    // The chain's two products per tile step, L_{k+1,k} and the update of B_{k+1,k+1}, have kernels of their own that cut
    // a 128 x 128 tile into 16 x 16 pieces (k_chain_l / k_chain_u: 3-5 us for one matrix, where the tile kernel needs 12).
    // With MANY matrices in lock-step -- a batch of evaluations (midn.hip): 192 weight matrices of 32 evaluations -- those
    // are throughput work, and kernels built for latency do it at 6-10 TF: 77 + 62 us of a 215 us tile step at N = 512.
    // (as tile tasks 49 + 31-72 us; N = 512, 32 evaluations 3 960 -> 4 090 /s, N = 497 5 930 -> 6 070; the diagonal blocks of
    // the next step, 4-5 x slower beside the step's updates than alone, are what such a step waits for now)
    const bool wide_chain = nbatch >= GPRN_WIDE_CHAIN;

    // stream3 at the start of step k: raise F_INNER of the step before, then wait for diag(k)
    auto side_sync = [&](int k) -> int {
        if (!use_flags) {
            if (inner_k >= 0) HIP_TRY(c, raise(s1, inner_k, F_INNER));
            inner_k = -1;
            HIP_TRY(c, await(s1, k, F_DIAG));
            return GPRN_OK;
        }
        const bool skip = inner_k >= 0 && withheld(F_INNER);
        unsigned* const up = inner_k >= 0 && !skip ? slot(inner_k, F_INNER) + 1 : nullptr;
        inner_k = -1;
        return flag_sync(s1, up, slot(k, F_DIAG) + 1);
    };
    // (never withheld by the test hook: this raise is also what the last tile step's wait consumes)
    auto flush_inner = [&]() -> int {              // nothing else follows on stream3 soon
        if (inner_k >= 0)
            HIP_TRY(c, use_flags ? hipStreamWriteValue32(s1, slot(inner_k, F_INNER) + 1, epoch, 0)
                                 : hipEventRecord(events[F_INNER], s1));
        inner_k = -1;
        return GPRN_OK;
    };
    // the panel launch's last workgroup holds it open until L_{k+1,k} is there (F_MINIL), so the in-panel updates
    // behind it need no stream wait (F_XW's counter word counts the workgroups; the flag itself is not raised)
    auto x_part_then = [&](int k) {
        return use_flags ? Signal{slot(k, F_XW), 0u, slot(k, F_MINIL) + 1, epoch, timed_out} : nosig;
    };
    // With one or two matrices stream3's synchronisation kernel is folded into the panel launch: F_INNER of the step
    // before goes up when its first workgroup runs, every workgroup waits for the diagonal block itself -- one launch
    // less per step on stream3 (config 2 737 -> 751 sweeps/s, config 3 +0.5 %; with six matrices, whose panel launches
    // are hundreds of workgroups that would all poll: -1 %, and beyond may_poll not safe)
    auto folds_sync = [&](int k) {
        if (k < 0 || k >= c->T) return false;
        const gprn_ctx::StepRange& sk = c->steps[set][k];
        return use_flags && nbatch <= 2 && sk.npanel_l > 0 && sk.npanel > 1 && may_poll(2 * (sk.npanel - 1) * (size_t)nbatch);
    };

    // B is still to be built (run_phase): only what the first outer panel's tile steps touch; its K = 512 update forms
    // the other tiles from K on the way in (bit 5 of their tasks; tile_mma ft_K) -- 16 N^2 bytes of HBM traffic per
    // matrix and three quarters of k_build_B's time at the head of the phase less
    bool ft_fused = false;
    if (c->build_pending) {
        const int pend = c->build_pending;
        c->build_pending = 0;
        const gprn_ctx::OuterRange& o0 = c->outers[set][0];
        const int outer = o0.k1 - o0.k0;
        // (the 64 x 64 tile kernel only: every launch of the first panel's update must use that shape)
        ft_fused = c->ft_s_phase && pend == nbatch && c->T > outer &&
                   shape_upd(o0.nfirst) == TS_64x64 && shape_upd(o0.nnext) == TS_64x64;
        if ((rc = vec_build_B(c, pend, s0, ft_fused ? 1 : 0, outer))) return rc;
    }
    if (!use_flags && c->chain_started) {          // event schedule: nothing to gate the caller's side work on
        std::function<int()> f;
        f.swap(c->chain_started);
        if ((rc = f())) return rc;
    }

    // Outer update of panel J (K = its width).  stream3, which has seen every tile of the panel: what the chain touches
    // first in the next panel ("first"); next stream: the rest of the next panel; bulk stream: everything beyond, look-ahead
    // part first.  The chain itself goes straight on with the next diagonal block.
    auto do_outer = [&](int J) -> int {
        const gprn_ctx::OuterRange& o = c->outers[set][(size_t)J];
        pending_outer = -1;
        if (use_flags) {
            // F_PANEL up, then the waits for the previous panel's "rest" (its look-ahead part when that is a launch of
            // its own) and "next" -- they wrote the tiles "first" updates -- in ONE kernel on stream3
            FlagOps ops = {{slot(J, F_PANEL) + 1, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
            if (rest_J >= 0) ops.wait[0] = slot(rest_J, sr ? F_RESTA : F_REST) + 1;
            if (next_J >= 0) { ops.wait[1] = slot(next_J, F_NEXT) + 1; next_J = -1; }
            if ((rc = flag_multi(s1, ops))) return rc;
        } else {
            HIP_TRY(c, raise(s1, J, F_PANEL));
            if (rest_J >= 0) HIP_TRY(c, await(s1, rest_J, sr ? F_RESTA : F_REST));
            if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
        }
        // the first panel's update forms the tiles of B it touches from K (run_phase built only the others)
        struct FtScope { gprn_ctx* c; ~FtScope() { c->ft_s_now = nullptr; } } ft_scope{c};
        c->ft_s_now = (o.k0 == 0 && ft_fused) ? c->ft_s_phase : nullptr;
        if (o.k1 < c->T) side_stamp(o.k1, 6);          // (stamps of the NEXT panel's first step: behind the waits, behind "first")
        if ((rc = tiles(o.first0, o.nfirst, s1, shape_upd(o.nfirst), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
        if (o.k1 < c->T) side_stamp(o.k1, 7);
        // F_FIRST: by the first workgroup of the next launch on stream3 when that is a panel launch with the
        // synchronisation folded in, else a stream write
        if (use_flags && folds_sync(o.k1)) pending_up = slot(J, F_FIRST) + 1;
        else HIP_TRY(c, raise(s1, J, F_FIRST));
        if (o.nfirst > 0) first_J = J;
        // "next": behind the panel (F_PANEL) and what wrote its tiles before (the previous "rest", look-ahead part)
        if (use_flags && sr && rest_J >= 0) {
            FlagOps ops = {{nullptr, nullptr}, {slot(J, F_PANEL) + 1, slot(rest_J, F_RESTA) + 1, nullptr, nullptr}};
            if ((rc = flag_multi(sn, ops))) return rc;
        } else {
            HIP_TRY(c, await(sn, J, F_PANEL));
            if (sr && rest_J >= 0) HIP_TRY(c, await(sn, rest_J, F_RESTA));
        }
        if ((rc = tiles(o.next0, o.nnext, sn, shape_upd(o.nnext), GPRN_T_PANEL, nosig, noaw, TG_NEXT))) return rc;
        // (a stream write, not the launch's own end-of-kernel signal: with a fence and an atomic at the end of each of its
        // several hundred workgroups 116.6 vs 117.6 sweeps/s at config 3 with two matrices, 113.4 with six)
        HIP_TRY(c, raise(sn, J, F_NEXT));
        if (o.nnext > 0) next_J = J;
        if (o.nrest) {
            // 64 x 64 workgroups: short-lived, they hand CUs to the other streams' kernels four times as often as the
            // 8-wave 128 x 128 form (109.0 vs 105.9 sweeps/s at config 3)
            if (sr) {
                HIP_TRY(c, await(s2, J, F_PANEL));
                if ((rc = tiles(o.rest0, o.nrestA, s2, TS_64x64, GPRN_T_UPDATE_AHEAD, nosig, noaw, TG_AHEAD))) return rc;
                // F_RESTA: by the first workgroup of the "bulk" launch behind it (gprn_ctx::start_flag_now) when there is one
                const bool resta_by_bulk = use_flags && o.nrest > o.nrestA;
                if (resta_by_bulk) { c->start_flag_now = slot(J, F_RESTA) + 1; c->start_value_now = epoch; }
                else HIP_TRY(c, raise(s2, J, F_RESTA));
                rc = tiles(o.rest0 + o.nrestA, o.nrest - o.nrestA, s2, TS_64x64, GPRN_T_UPDATE, nosig, noaw, TG_BULK);
                c->start_flag_now = nullptr;
                if (rc) return rc;
            } else if ((rc = tiles(o.rest0, o.nrest, s2, TS_64x64, GPRN_T_UPDATE, nosig, noaw, TG_BULK))) return rc;
            HIP_TRY(c, raise(s2, J, F_REST));
            rest_J = J;
        }
        // rows [k0, k1) of X are final once stream3 is through with the panel: their share of the phase's O(N^2)
        // reductions goes behind the panel's bulk update on the bulk stream (run_phase, api_sweep.hip)
        if (c->rows_final) {
            if (!(o.nrest && sr) && sn != s2) HIP_TRY(c, await(s2, J, F_PANEL));
            if ((rc = c->rows_final(o.k0, o.k1, s2))) return rc;
            c->rows_done = o.k1;
            tail_on_s2 = true;
        }
        return GPRN_OK;
    };

    for (size_t J = 0; J < c->outers[set].size(); ++J) {
        const gprn_ctx::OuterRange& o = c->outers[set][J];
        for (int k = o.k0; k < o.k1; ++k) {
            const gprn_ctx::StepRange& s = c->steps[set][k];
            if (pending_outer >= 0 && s.npanel_l == 0 && (rc = do_outer(pending_outer))) return rc;
            // ---- the chain
            if ((rc = launch_diag(c, c->d_ptrs, nbatch, c->ld, k, c->d_info_cur, s0, in_kernel(k, F_DIAG), noaw))) return rc;
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
                // last tile step of the matrix: row k of the inverse is all that is left.  Its workgroups wait for
                // stream3's last update themselves when they are few (a stream wait is a 5 us kernel of its own, and the
                // flag is up or about to be when they start); a one-wave kernel waits for them when they are many
                if ((rc = flush_inner())) return rc;
                if (k > 0 && may_poll(2 * s.npanel * (size_t)nbatch)) {
                    if ((rc = tiles(s.panel0, s.npanel, s0, TS_128x64_ATRI, GPRN_T_PANEL, nosig, in_kernel_wait(k - 1, F_INNER), TG_PANEL))) return rc;
                    continue;
                }
                if (k > 0 && use_flags) { if ((rc = flag_sync(s0, nullptr, slot(k - 1, F_INNER) + 1))) return rc; }
                else if (k > 0) HIP_TRY(c, await(s0, k - 1, F_INNER));
                if ((rc = tiles(s.panel0, s.npanel, s0, TS_128x64_ATRI, GPRN_T_PANEL, nosig, noaw, TG_PANEL))) return rc;
                continue;
            }
            // L_{k+1,k} reads what stream3's in-panel update of step k-1 wrote: with one or two matrices (the chain
            // bounds the phase) its 8 workgroups per matrix poll that flag themselves; with more, the phase is bound by
            // the tile kernels' throughput and 8 x batch resident 512-thread workgroups that only poll keep bulk
            // workgroups off their CUs: a one-wave kernel waits instead (+0.8 % at config 3, +2 % at config 4)
            const bool spin = use_flags && k > 0 && nbatch <= 2;
            if (use_flags && k > 0 && !spin && (rc = flag_sync(s0, nullptr, slot(k - 1, F_INNER) + 1))) return rc;
            if (k > 0 && !use_flags) HIP_TRY(c, await(s0, k - 1, F_INNER));
            if (wide_chain) {
                // the same two products as tasks of the tile kernel: the step's first panel task and first update task
                // (ensure_tasks), a few hundred 64-row workgroups at its K = 128 rate
                if ((rc = tiles(s.panel0, 1, s0, TS_64x128_BTRI, GPRN_T_PANEL, nosig, noaw, TG_PANEL))) return rc;
                if (use_flags) { c->start_flag_now = slot(k, F_MINIL) + 1; c->start_value_now = epoch; }
                else HIP_TRY(c, raise(s0, k, F_MINIL));
                rc = tiles(s.upd0, 1, s0, TS_64x64, GPRN_T_PANEL, nosig, noaw, TG_INNER);
                c->start_flag_now = nullptr;
                if (rc) return rc;
            } else {
            // (a prior matrix: the tile by substitution, the panel kernel's ACC form on the step's first task)
            if (c->acc_now) rc = launch_panel(c, c->d_tasks + s.panel0, 1, 0, c->d_ptrs, nbatch, c->ld, s0, nosig,
                                              spin ? in_kernel_wait(k - 1, F_INNER) : noaw);
            else rc = launch_tile_rows(c, k, c->d_ptrs, nbatch, c->ld, 0, GPRN_T_PANEL, s0, nosig,
                                       spin ? in_kernel_wait(k - 1, F_INNER) : noaw);
            if (rc) return rc;
            if (!use_flags) HIP_TRY(c, raise(s0, k, F_MINIL));
            // (flag schedule: L_{k+1,k}'s flag goes up at the START of the update launch behind it on the chain stream
            // instead of at the end of its own -- 1.7 us less between the two at every tile step)
            if ((rc = launch_tile_rows(c, k, c->d_ptrs, nbatch, c->ld, 1, GPRN_T_PANEL, s0, nosig, noaw,
                                       use_flags ? slot(k, F_MINIL) + 1 : (unsigned*)nullptr, epoch))) return rc;
            }
            // The outer update of the previous panel is ENQUEUED here, behind the chain's three launches of this panel's
            // first step: its dozen stream operations and launches take the host 60-100 us, during which the chain stream
            // ran dry at every panel boundary (profiles/r02_chain_timeline_cfg3.txt)
            if (pending_outer >= 0 && (rc = do_outer(pending_outer))) return rc;
            // ---- stream3: the rest of the panel, then the rest of the in-panel updates
            side_stamp(k, 0);
            const bool fold_sync = folds_sync(k);
            if (!fold_sync) {
                if (pending_up) { HIP_TRY(c, hipStreamWriteValue32(s1, pending_up, epoch, 0)); pending_up = nullptr; }
                if ((rc = side_sync(k))) return rc;
            }
            side_stamp(k, 1);
            if (use_flags) {
                unsigned *up = nullptr, *up2 = nullptr;
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
            } else {
                if ((rc = tiles(s.panel0 + 1, s.npanel_l - 1, s1, TS_64x128_BTRI, GPRN_T_PANEL, nosig, noaw, TG_PANEL))) return rc;
                if ((rc = tiles(s.panel0 + s.npanel_l, s.npanel - s.npanel_l, s1, TS_128x64_ATRI, GPRN_T_PANEL, nosig, noaw, TG_PANEL))) return rc;
                HIP_TRY(c, await(s1, k, F_MINIL));
            }
            side_stamp(k, 2);                              // behind the panel
            // At the FIRST step of a panel the step's updates of the panel's other columns have to wait for the previous
            // panel's outer update ("next"), but the two tiles the chain's next step needs do not: they are the next
            // panel's eager tiles, kept up to date step by step and left out of the outer update (ensure_tasks).  With one
            // or two matrices those two go first, in a launch of their own that raises F_INNER itself, BEFORE the wait for
            // "next" (config 2: +3 %; with six matrices the phase is bound by throughput and the extra launch costs 1 %;
            // at EVERY step it is slower: one more launch per step on stream3, itself a serial chain of launches).
            const size_t ncrit = s.ncol1 > 0 ? s.ncol1 - 1 : 0;
            if (use_flags && ncrit > 0 && next_J >= 0 && nbatch <= 2) {
                const bool skip = withheld(F_INNER);
                if ((rc = tiles(s.upd0 + 1, ncrit, s1, TS_64x64, GPRN_T_PANEL, skip ? nosig : in_kernel(k, F_INNER)))) return rc;
                side_stamp(k, 3);
                HIP_TRY(c, await(s1, next_J, F_NEXT));
                next_J = -1;
                side_stamp(k, 4);
                if ((rc = tiles(s.upd0 + 1 + ncrit, s.nupd - 1 - ncrit, s1, shape_upd(s.nupd - 1 - ncrit)))) return rc;
                side_stamp(k, 5);
            } else {
                if (next_J >= 0) { HIP_TRY(c, await(s1, next_J, F_NEXT)); next_J = -1; }
                side_stamp(k, 4);
                if ((rc = tiles(s.upd0 + 1, s.nupd - 1, s1, shape_upd(s.nupd - 1)))) return rc;
                side_stamp(k, 5);
                // (hundreds of workgroups: a fence + atomic in each would cost more than one stream write; the flag goes
                // up with stream3's next synchronisation)
                if (use_flags) inner_k = k;
                else HIP_TRY(c, raise(s1, k, F_INNER));    // an event wait sees only records made before it: the
                                                           // chain's wait for step k is enqueued at step k + 1
            }
        }
        if ((rc = flush_inner())) return rc;           // the chain's next step must not queue behind the outer update
        if (o.nfirst + o.nnext + o.nrest == 0) continue;
        pending_outer = (int)J;
    }
    if (pending_outer >= 0 && (rc = do_outer(pending_outer))) return rc;
    if (pending_up) { HIP_TRY(c, hipStreamWriteValue32(s1, pending_up, epoch, 0)); pending_up = nullptr; }
    if (tail_on_s2) HIP_TRY(c, raise(s2, 0, F_TAIL));
    if (use_flags) {
        // the chain stream joins the others in ONE kernel (bounded by the budget like every in-kernel wait)
        FlagOps ops = {{nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
        int nw = 0;
        if (first_J >= 0) ops.wait[nw++] = slot(first_J, F_FIRST) + 1;
        if (next_J >= 0) ops.wait[nw++] = slot(next_J, F_NEXT) + 1;
        if (rest_J >= 0) ops.wait[nw++] = slot(rest_J, F_REST) + 1;
        if (tail_on_s2) ops.wait[nw++] = slot(0, F_TAIL) + 1;
        return nw ? flag_multi(s0, ops) : GPRN_OK;
    }
    if (first_J >= 0) HIP_TRY(c, await(s0, first_J, F_FIRST));
    if (next_J >= 0) HIP_TRY(c, await(s0, next_J, F_NEXT));
    if (rest_J >= 0) HIP_TRY(c, await(s0, rest_J, F_REST));
    if (tail_on_s2) HIP_TRY(c, await(s0, 0, F_TAIL));
    return GPRN_OK;
}

// a dependency wait inside a chain kernel gave up (see Await): the results of that call are void
int factor_check_waits(gprn_ctx* c)
{
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
    if (!c->d_sig) return GPRN_OK;
    unsigned word[3] = {0, 0, 0};                      // sticky word, budget, which flag (spin_until)
    HIP_TRY(c, hipMemcpy(word, c->d_sig + (size_t)c->sig_T * GPRN_FLAG_KINDS * 2, sizeof(word), hipMemcpyDeviceToHost));
    if (word[0]) {
        hipMemset(c->d_sig + (size_t)c->sig_T * GPRN_FLAG_KINDS * 2, 0, sizeof(unsigned));
        static const char* const kind_name[GPRN_FLAG_KINDS] = {"DIAG", "MINIL", "INNER", "PANEL", "NEXT", "REST", "FIRST", "XW", "RESTA", "TAIL"};
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


int factor_invert(gprn_ctx* c, int nbatch, bool prior)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    struct AccScope { gprn_ctx* c; ~AccScope() { c->acc_now = false; } } acc_scope{c};
    c->acc_now = c->acc_opt < 0 ? prior : c->acc_opt != 0;
    static int step_stamps_env = -1;               // GPRN_STEP_STAMPS=1/2 (probes): in-kernel clock stamps of the chain / of stream3 too
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
    // latency set of task lists for problems with little work in total (measured: +11 % at N = 2048 x 1 matrix,
    // -3 % at N = 4096 x 2)
    rc = factor_invert_launches(c, nbatch, nbatch * c->T <= GPRN_LAT_MAX ? 1 : 0);
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

// BUF_B of every slot = lower(X^T X), X in BUF_X (L in BUF_B is overwritten)
int lauum_lower(gprn_ctx* c, int nbatch, hipStream_t stream)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    return launch_tiles(c, c->d_tasks + c->lauum0, c->nlauum, c->d_ptrs, nbatch, c->ld,
                        GPRN_T_LAUUM, stream);
}
