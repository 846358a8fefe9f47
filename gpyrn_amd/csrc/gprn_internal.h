// Internal declarations shared by the HIP translation units of libgprn_hip.so.
// Public C ABI: include/gprn_hip.h.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/gprn_hip.h"

#define GPRN_TILE 128          // tile edge of the blocked factorisation (nb)
#define GPRN_KC 16             // K chunk staged through LDS per pipeline stage
#define GPRN_NBUF 4            // per-GP buffer slots addressable by a tile task
#define GPRN_OUTER 4           // tiles per outer panel: bulk updates contract over 4*128 = 512
#define GPRN_OUTER_SMALL 16    // ... when batch x tiles <= 32 (latency-bound: measured +11 % at N=2048, batch 1)
#define GPRN_LAT_MAX 32        // batch x tiles up to which a factorisation runs on the latency set of task lists
#define GPRN_FEW_TASKS 4000    // tasks x batch above which a tile launch uses 128 x 128 workgroups (100 ... 8000 swept)
#ifndef GPRN_WIDE_CHAIN
#define GPRN_WIDE_CHAIN 48     // matrices in lock-step from which the chain's two products per tile step run on the tile kernel
#endif
#define GPRN_XCD_CHUNK_LOG2 4  // consecutive task-list entries that meet in one XCD's L2 (k_tile_gemm): 16

// The hand-over of k_reduce_finalize (vecops.hip) without fences relies on what gfx942 / gfx950 do with agent-scope
// stores and on vmcnt counting store acknowledgements; any other target gets the release / acquire form.
#if defined(__gfx942__) || defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define GPRN_RELAXED_HANDOVER 1
#else
#define GPRN_RELAXED_HANDOVER 0
#endif

// Pointers fetched from a device pointer table are generic to the compiler, which then emits
// FLAT loads; those also tick the LDS counter (lgkmcnt), so the wait before the first MFMA of
// a K-chunk would drain the global prefetch of the next chunk.  Casting to the global
// address space yields global_load/global_store (vmcnt only) and keeps the prefetch in flight.
#define GPRN_GLOBAL __attribute__((address_space(1)))
typedef GPRN_GLOBAL double* gptr_t;
typedef const GPRN_GLOBAL double* gcptr_t;

// buffer slots of a tile task (index into the per-GP pointer table)
enum { BUF_B = 0, BUF_X = 1, BUF_K = 2, BUF_KLINV = 3 };
// c_mode of a tile task
enum { CM_SET = 0, CM_SUB = 1, CM_SETNEG = 2 };

// One 128x128 output tile of  C (op)= A . B  over klen, all operands tiles of
// square row-major matrices with leading dimension ld.
//   a_mode 0: A element (m,k) at a_off + m*ld + k     (k contiguous)
//   a_mode 1: A element (m,k) at a_off + k*ld + m     (m contiguous, i.e. A^T stored)
//   b_mode 0: B element (k,n) at b_off + n*ld + k     (k contiguous, "NT")
//   b_mode 1: B element (k,n) at b_off + k*ld + n     (n contiguous, "NN")
struct TileTask {
    int64_t c_off, a_off, b_off;
    int32_t klen;
    uint8_t c_buf, a_buf, b_buf;
    uint8_t modes;             // bits 0-1 c_mode, bit 2 a_mode, bit 3 b_mode, bit 4: symmetric update of a diagonal tile
                               // (C -= A A^T, same operand twice): only the lower triangle of the result is ever read;
                               // bit 5: the first K = 512 update of a tile of B (first outer panel): may form it from K
};
static inline uint8_t tile_modes(int c_mode, int a_mode, int b_mode, int lower_only = 0) {
    return (uint8_t)((c_mode & 3) | ((a_mode & 1) << 2) | ((b_mode & 1) << 3) | ((lower_only & 1) << 4));
}

#define HIP_TRY(ctx, expr)                                                     \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);    \
            return GPRN_E_HIP;                                                 \
        }                                                                      \
    } while (0)

struct Profiler {
    bool on = false;
    struct Rec { int fam; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double ms[GPRN_T_COUNT] = {0};
    int64_t n[GPRN_T_COUNT] = {0};
};

struct KernelSpec {            // how latent GP g gets its K
    bool set = false, uploaded = false;
    int n_ops = 0, n_params = 0, nugget = 0;
    int32_t ops[3 * GPRN_MAX_OPS];
    double params[GPRN_MAX_KPARAMS];
};

// Several evaluations of ONE problem side by side (midn.hip: an optimiser's simplex, emcee's walkers): the slots of a phase
// then belong to different evaluations, each with its own copy of the per-problem arrays.  slot_eval: slot -> evaluation
// (null: one evaluation, every stride unused); the strides are doubles between two evaluations' copies.
struct EvalMap {
    const int* slot_eval;
    size_t state;              // mu, var: (p + 1) q N
    size_t yv;                 // y - mean, variance: p N
    size_t scal;               // per-GP scalars of a sweep: 3 G + q q
    size_t G;                  // log det K: G
};

struct DeviceStreams {         // one per device and process, see gprn_create
    hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
    int device = 0, refs = 0;
    int use_flags = -1;        // the flag schedule's verdict for these streams (factor_use_flags), -1: not probed
    std::recursive_mutex mu;   // held for the length of every entry point
};

struct gprn_ctx {
    int device = 0;
    DeviceStreams* shared = nullptr;
    hipStream_t stream = nullptr;    // everything, incl. the latency chain of the factorisation
    hipStream_t stream2 = nullptr;   // bulk trailing updates running behind the chain (look-ahead)
    hipStream_t stream3 = nullptr;   // in-panel work that is off the chain (panel rest, inner rest)
    hipStream_t stream4 = nullptr;   // the next panel's share of an outer update ("next"), beside the previous panel's "rest"
    hipEvent_t ev_panel = nullptr, ev_rest = nullptr, ev_next = nullptr, ev_nodes = nullptr, ev_q1 = nullptr, ev_resta = nullptr;
    hipEvent_t ev_diag = nullptr, ev_minil = nullptr, ev_inner = nullptr, ev_first = nullptr;
    // head / tail of a phase beside its factorisation (run_phase, api_sweep.hip; factor_invert_split, factor.hip)
    hipEvent_t ev_tail = nullptr;
    bool node_term_done = false;     // the node phase's mu^T K^-1 mu went to the bulk stream beside the weight phase (run_phase)
    // run_phase: called (by schedules that know) once tile rows [r0, r1) of X are final in `stream` order -- the O(N^2)
    // reductions over X's rows (X z, column norms, X^T u) then run beside the rest of the factorisation instead of
    // behind it; rows_done: first tile row the caller still has to do itself
    std::function<int(int r0, int r1, hipStream_t stream)> rows_final;
    int rows_done = 0;
    hipStream_t prof_stream = nullptr;
    std::string err;
    int info_gp = -1;
    Profiler prof;
    int prof_mask = 0;               // bit per GPRN_T_* family
    bool prof_open = false;

    // ---- problem
    int N = 0, p = 0, q = 0, G = 0, ld = 0, T = 0;   // ld = N padded to GPRN_TILE, T = ld/TILE
    double *d_time = nullptr, *d_yraw = nullptr, *d_yerr2 = nullptr;
    double *d_yres = nullptr, *d_variance = nullptr;
    std::vector<double> h_yerr2;
    double *d_mu = nullptr, *d_var = nullptr;        // (p+1, q, N) each, reference layout
    double *d_mu_save = nullptr, *d_var_save = nullptr;
    bool have_yres = false, have_jit = false, have_muvar = false, factored = false;

    // ---- sharding
    int world = 1, rank = 0;
    std::vector<int> owner;          // G entries (empty until known when world > 1)
    std::vector<int> loc_nodes, loc_weights;   // latent GPs of this rank, ascending = slot order
    void* comm = nullptr;            // ncclComm_t
    void* shm = nullptr;             // ShmComm: rehearsal transport of one-GPU boxes (api.hip)
    double* d_agree = nullptr;       // one word: did any rank's call time out (with_event_fallback, api_internal.h)
    void* watch = nullptr;           // WatchEntry (api.hip): this context's slot of the collective watchdog, while it has a communicator
    int comm_budget_s = -1;          // gprn_set_option "comm_budget_s"; -1: GPRN_COMM_BUDGET_S or 600

    // ---- per latent GP, persistent across sweeps (only for GPs this rank needs)
    std::vector<KernelSpec> kspec;   // G
    std::vector<double*> K;          // G   (ld x ld), null when not held
    std::vector<double*> KLinv;      // G   chol(K)^-1 lower
    std::vector<double*> Kinv;       // q   K_j^-1 lower, only for nodes j>=1 that feed quirk Q1
    std::vector<double*> Sig;        // G   explicit Sigma of the last sweep (keep_sigma only)
    bool keep_sigma = false;
    bool q1_pending = false;
    double* d_logdetK = nullptr;     // G
    // ---- workspaces: nslot pairs (B, X), nslot = max local GPs of a phase
    int nslot = 0;                   // local nodes + local weights: every local GP has its own (B, X)
    int slot0 = 0;                   // first slot of the running phase (0 nodes, #local nodes weights)
    std::vector<double*> wsB, wsX;
    double** d_ptrs = nullptr;       // the table the launchers use right now (one of the three below)
    // host copies of the live pointer tables (device address -> rows): the chain's kernels take the few pointers
    // they need as kernel arguments instead of fetching them from the table (one memory round trip less on the
    // critical path of every tile step); tab_note / tab_forget / tab_rows, factor.hip
    std::vector<std::pair<double**, std::vector<double*>>> tab_host;
    double **tab_node = nullptr, **tab_weight = nullptr, **tab_setup = nullptr;  // [nslot][GPRN_NBUF]
    int *d_slotgp_node = nullptr, *d_slotgp_weight = nullptr, *d_slotgp_setup = nullptr;
    bool tables_ready = false;
    // per-slot vectors (ld each): d, s, pred, w(=K pred), z, u, colsq, colt
    double *d_d = nullptr, *d_s = nullptr, *d_pred = nullptr, *d_w = nullptr,
           *d_z = nullptr, *d_u = nullptr, *d_cs = nullptr, *d_ct = nullptr;
    double* d_part = nullptr;        // partial column sums scratch [nslot][T][2][ld]
    double* d_fin_terms = nullptr;   // k_reduce_finalize: per-element terms of tr B^-1 and log det B [nslot][2][ld]
    unsigned* d_fin_tickets = nullptr;   // ... and its per-slot ticket counters (zero between launches)
    // per-GP scalars of the running sweep, one allocation (all-reduced as one message):
    // logdetB[G], trBinv[G], muKmu[G], Q1 traces [q*q]
    double* d_scal = nullptr;        // (two copies: a sweep's ELBO assembly may run beside the next sweep, sweep_impl)
    double* d_scal_base = nullptr;
    double* d_elbo_part = nullptr;   // scratch of the ELBO assembly, two copies
    double *d_logdetB = nullptr, *d_trBinv = nullptr, *d_muKmu = nullptr, *d_q1 = nullptr;
    double* d_out = nullptr;         // per sweep: elbo, logl, logp, ent
    int out_cap = 0;
    int* d_info = nullptr;           // [3][nslot] first failing pivot per slot: setup, node phase, weight phase
    int* d_info_cur = nullptr;       // the row factor_invert writes to
    // prediction scratch (gprn_predict): K* and (X K*^T)^T per local GP, [ns_pad x ld] each
    std::vector<double*> predKs, predWT;
    // host-evaluated matrices staged for the next gprn_predict (gprn_predict_upload): K + 1.25e-12 I (N x N), K* (ns x N), k** (ns)
    struct PredStage { int ns = 0; std::vector<double> K, Kstar, kss; };
    std::map<int, PredStage> pred_stage;
    size_t pred_cap = 0;
    double **tab_pred = nullptr;
    int* d_slotgp_all = nullptr;
    // scratch of the diagnostic entry points
    double* d_test[3] = {nullptr, nullptr, nullptr};
    size_t test_cap[3] = {0, 0, 0};
    // tile-task lists for the factorisation at the current T (device)
    TileTask* d_tasks = nullptr;
    unsigned* d_sig = nullptr;       // completion signals of the chain: (tile step, kind) -> {counter, flag}
    int sig_T = 0;
    unsigned epoch = 0;              // value the flags take in the current factor_invert call
    // How cross-stream dependencies of the factorisation travel (factor.hip): 1 = 32-bit flags in device
    // memory (stream memory operations + in-kernel waits), 0 = HIP events, -1 = not decided yet.  Decided per
    // context from the device and the environment; latched to 0 after an in-kernel wait timed out.
    int use_flags = -1;
    // Panel steps by substitution instead of products with explicit inverses (diag_tile.h ACC): acc_now is what the running
    // factor_invert uses; option "accurate_factor": -1 every factorisation of a PRIOR matrix (set-up, prediction, prior draws),
    // 0 never, 1 always (the sweeps' B too: diagnostics)
    bool acc_now = false;
    int acc_opt = -1;
    int fenced_finalize = 0;         // gprn_set_option "fenced_finalize" (tests): k_reduce_finalize's release / acquire form
    int wait_budget_ms = 2000;       // wall-clock budget of one in-kernel wait (gprn_set_option "wait_budget_ms")
    int withhold_inner = 0;          // test hook: the n-th F_INNER raise of the next call is skipped (0 = none)
    int fallbacks = 0;               // calls that were re-run on the event schedule after a time-out
    std::string last_timeout;        // which flag the last time-out was waiting for (factor_check_waits)
    // LDS pads of the tile launches (gemm_tile.hip launch_tiles), KiB; -1: the environment's / the default
    int pad_kb_opt = -1, pad_small_kb_opt = -1;
    int sig_budget_ms = -1;          // budget the device word holds
    // enqueued by the next factor_invert behind its first diagonal block (factor.hip)
    std::function<int()> chain_started;
    // GPRN_STEP_STAMPS=1 (probes): per tile step and chain kernel (diag, L, U) the 100 MHz clock at its start, after its
    // wait and at its end -- the launch schedule's chain as it really ran (a kernel trace slows the chain's small
    // kernels by 15 %); [phase slot][T][3 kernels][3 stamps], printed by factor_check_waits
    unsigned* start_flag_now = nullptr;            // launch_tiles: a flag word the next tile launch sets to start_value_now when its
    unsigned start_value_now = 0;                  // first workgroup runs (the flag of the launch BEFORE it on its stream)
    unsigned long long* d_step_stamps = nullptr;
    int step_stamps_T = 0, step_stamps_n = 0;
    int step_stamps_batch[8] = {0};
    unsigned long long *d_side_stamps = nullptr, *side_stamps = nullptr;   // GPRN_STEP_STAMPS=2: stream3's clock, [T][8]
    int side_stamps_ph = -1;
    size_t tasks_cap = 0;
    std::vector<TileTask> h_tasks;
    struct StepRange { size_t panel0, npanel_l, npanel, upd0, nupd, ncol1; };   // per tile step: panel (L part first, then X part), in-panel update (the first ncol1 tasks: the chain's own tile and the two its next step touches)
    // two sets: [0] throughput schedule (outer panel = GPRN_OUTER tiles), [1] latency schedule for
    // small problems (batch x tiles <= 32; wider outer panels: fewer bulk-update joins on the chain)
    std::vector<StepRange> steps[2]; // T entries each
    // per outer panel: the three parts of its K = (k1 - k0) * 128 update -- "first" (the next panel's first column of B and
    // first row of R), "next" (the rest of the next panel's columns / rows), "rest" (everything beyond; its first nrestA
    // tasks are what the NEXT panel's outer update writes again)
    struct OuterRange { int k0, k1; size_t first0, nfirst, next0, nnext, rest0, nrest, nrestA; };
    std::vector<OuterRange> outers[2];
    size_t lauum0 = 0, nlauum = 0;
    int tasks_T = 0;
    // run_phase -> factor_invert_split: s = sqrt(d) of the phase's slots when B's tiles beyond the first outer panel are
    // still to be formed -- by the first panel's K = 512 update, on the way in (tile_mma ft_K); ft_s_now: what launch_tiles
    // passes to the kernel right now (set around that update's launches only)
    int build_pending = 0;           // run_phase: B of this many slots is still to be built by the next factor_invert
    const double* ft_s_phase = nullptr;
    const double* ft_s_now = nullptr;
    int overlap_opt = -1;            // gprn_set_option "overlap" (api_sweep.hip overlap_mask); -1: the default
    // ---- small-N path (smalln.hip): problems of one or two tiles run a half-sweep as ONE launch, one workgroup per latent GP
    int small_opt = -1;              // gprn_set_option "small_path": 0 never, else wherever it applies (small_applies)
    double** d_kinv_tab = nullptr;   // [q] device pointers K_j^-1 (quirk Q1), for k_small_tail
    double** d_kinv_out = nullptr;   // [nslot] per set-up job: where k_small_prior puts K^-1, or null
    unsigned* d_small_ticket = nullptr;
    unsigned long long* d_small_stamps = nullptr;   // GPRN_SMALL_STAMPS (probes): stage clocks of the node half-sweep's workgroup 0
    double *d_mu_alt = nullptr, *d_var_alt = nullptr;   // the second copy of the state (smalln.hip: a sweep reads one, writes the other)
    int* d_loop_ctl = nullptr;       // gprn_elbocalc on the small path: [0] done, [1] iterNumber, [2] converged (+ pad)
    double* d_loop_hist = nullptr;   // ... the batch's ELBO values, then the loop's last three
    double *h_pin_in = nullptr, *h_pin_out = nullptr;    // pinned staging of gprn_elbocalc's inputs / read-backs
    void* small_batch = nullptr;     // SmallBatchMem (smalln.hip): buffers of gprn_elbocalc_batch
    size_t pin_in_cap = 0, pin_out_cap = 0;
    bool small_tabs_ready = false;   // the set-up's tables for this problem are on the device (factor_priors_small)
    bool small_sweep_ready = false;  // ... and what a sweep of the small path reads beside the phase tables (ensure_small_sweep_tabs)
    bool setup1_ready = false;       // tab_setup / d_slotgp_setup / tab_kinv1 hold the unsharded launch-path set-up's rows (factor_priors_single)
    double** tab_kinv1 = nullptr;    // [q - 1][GPRN_NBUF]: BUF_B = K_j^-1, BUF_X = chol(K_j)^-1, nodes j >= 1 (one X^T X launch)
    // ---- many evaluations side by side above one tile (midn.hip): a worker context holds the matrices and states of a
    // chunk of evaluations; its kernels find an evaluation's arrays through `ev`
    EvalMap ev = {nullptr, 0, 0, 0, 0};
    void* mid_batch = nullptr;       // MidBatch (midn.hip): the worker context and its slabs, owned by the PARENT context
    int batch_mem_mb = -1;           // gprn_set_option "batch_mem_mb": device memory one chunk of evaluations may take; -1: a share of what is free
    int last_batch_chunk = 0;        // read-only option "batch_chunk": evaluations per chunk in the last gprn_elbocalc_batch call
};

struct DeviceLock {                                // no-op for a null context (the entry point rejects it next)
    std::unique_lock<std::recursive_mutex> l;
    explicit DeviceLock(const gprn_ctx* c) { if (c && c->shared) l = std::unique_lock<std::recursive_mutex>(c->shared->mu); }
};


// ---- launchers (each enqueues on ctx->stream; no sync) ----
void prof_begin(gprn_ctx* c, int fam, hipStream_t stream = nullptr);   // nullptr = ctx->stream
void prof_end(gprn_ctx* c);

int launch_fill(gprn_ctx* c, const KernelSpec& ks, double* K, double nugget_val = 1e-6,
                const double* diag_add = nullptr);
// many small matrices in one launch (fill.hip; gprn_elbocalc_batch)
size_t fill_program_bytes();
bool fill_program_with(const KernelSpec& ks, const double* params, void* dst);
int launch_fill_batch(gprn_ctx* c, const void* d_programs, double* const* d_Ks, int n_matrices,
                      double* const* d_K2s = nullptr);      // d_K2s: a second copy of every matrix, or null
int launch_fill_rect(gprn_ctx* c, const KernelSpec& ks, double nugget_val, const double* d_tstar,
                     int ns, int ns_pad, double* Ks, double* kss);
// workgroup output shape of a tile launch (csrc/gemm_tile.hip)
enum { TS_128x128 = 0, TS_64x64 = 1, TS_64x128 = 2, TS_128x64 = 3,
       TS_64x128_BTRI = 4, TS_128x64_ATRI = 5 };   // panel products with the triangular X_kk (gemm_tile.hip TRI)
// launch family of a tile launch: a template tag of k_tile_gemm, so that a kernel trace reports every
// family under its own kernel name (panel products, in-panel K=128 updates, next-panel K=512 updates,
// bulk K=512 updates, everything else)
enum { TG_PANEL = 0, TG_INNER = 1, TG_NEXT = 2, TG_BULK = 3, TG_MISC = 4, TG_AHEAD = 5 };
// Completion signal of a launch, raised from the device: slot[0] counts the workgroups that have
// finished, the last one resets it and stores `value` to slot[1] (system scope).  Another stream
// picks it up with hipStreamWaitValue32 about 2 us later (profiles/probes/streamvalue.hip) -- no event
// record packet behind the kernel, no event wait packet on the consumer.
struct Signal {
    unsigned* slot; unsigned value;                // value 0: count the workgroups only, raise nothing
    // optionally the last workgroup then holds the launch open until *then_wait >= then_value: the
    // next launch of the stream starts behind that flag without a stream wait of its own
    const unsigned* then_wait; unsigned then_value; unsigned* timed_out;
};
// The other direction, for launches of a FEW workgroups only (a spinning launch that fills the GPU
// could keep its own producer from being dispatched): every workgroup of the launch waits at its
// start until *flag >= value; a wait that times out (about a second) sets *timed_out and goes on.
struct Await { const unsigned* flag; unsigned value; unsigned* timed_out; };
size_t lds_limit(int device);         // LDS bytes one workgroup may ask for, static + dynamic (gemm_tile.hip)
int launch_tiles(gprn_ctx* c, const TileTask* d_tasks, size_t ntasks, double** d_ptrs,
                 int nbatch, int ld, int fam, hipStream_t stream = nullptr, int shape = TS_128x128,
                 Signal sig = Signal{nullptr, 0, nullptr, 0, nullptr}, Await aw = Await{nullptr, 0, nullptr},
                 int tag = TG_MISC);
// < 1/2 (P - Kinv + a a^T), dK/dtheta_l > for every parameter of a kernel program by central differences of the
// program, on the device (fill.hip); out: n_params doubles of device memory, part: N doubles of scratch
int launch_grad_fd(gprn_ctx* c, const KernelSpec& ks, const double* Kinv, const double* P, const double* a,
                   double* part, double* out);
// the L part (n_l tasks) and the X part (n_x tasks) of a tile step's panel in one launch (gemm_tile.hip)
int launch_panel(gprn_ctx* c, const TileTask* d_tasks, size_t n_l, size_t n_x, double** d_ptrs, int nbatch, int ld,
                 hipStream_t stream, Signal sig, Await aw = Await{nullptr, 0, nullptr},
                 unsigned* raise_at_start = nullptr, unsigned raise_value = 0, unsigned* raise_at_start2 = nullptr);
// BUF_B and BUF_X of up to GPRN_ARG_SLOTS matrices as a kernel argument
#define GPRN_ARG_SLOTS 16
struct PtrArgs { double* p[GPRN_ARG_SLOTS][2];
                 unsigned long long* stamps; };   // GPRN_STEP_STAMPS: 100 MHz clock stamps of this launch (3 words), or null
void tab_note(gprn_ctx* c, double** d_tab, double* const* rows, size_t count);
void tab_forget(gprn_ctx* c, double** d_tab);                    // d_tab null: all of them
// rows of `nbatch` matrices starting at d_ptrs if a host copy is known (and nbatch fits), else false
bool tab_rows(gprn_ctx* c, double** d_ptrs, int nbatch, PtrArgs* out);
unsigned long long* step_stamp_ptr(gprn_ctx* c, int k, int which);   // GPRN_STEP_STAMPS (factor.hip)

// the chain's two products of a tile step at 16 x 16 granularity (gemm_tile.hip); mode 0: L_{k+1,k} in place, 1: the
// update of B_{k+1,k+1}
int launch_tile_rows(gprn_ctx* c, int k, double** d_ptrs, int nbatch, int ld, int mode, int fam,
                     hipStream_t stream, Signal sig, Await aw, unsigned* raise_at_start = nullptr, unsigned raise_value = 0);
int launch_diag(gprn_ctx* c, double** d_ptrs, int nbatch, int ld, int kblk, int* d_info,
                hipStream_t stream = nullptr, Signal sig = Signal{nullptr, 0, nullptr, 0, nullptr},
                Await aw = Await{nullptr, 0, nullptr});

#ifdef __HIPCC__
__device__ __forceinline__ size_t ev_of(const EvalMap& e, int slot) { return e.slot_eval ? (size_t)e.slot_eval[slot] : 0; }

// Spin of ONE thread until *flag >= value.  timed_out[0] is the sticky "a wait gave up" word of the call,
// timed_out[1] the budget of one wait in ticks of the 100 MHz constant clock (s_memrealtime): a wall-clock
// bound, not a spin count -- on a shared device a legitimate wait can be long.  Once any wait of the call
// has given up the others return at once (the results are void anyway; the host re-runs the call on events).
// Relaxed polling and ONE acquire after the match (acquire loads in the loop cost 2-3x per hop).
__device__ __forceinline__ void spin_until(const unsigned* flag, unsigned value, unsigned* timed_out)
{
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long budget = timed_out ? (unsigned long long)timed_out[1] : 200000000ull;
        for (;;) {
            __builtin_amdgcn_s_sleep(8);
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= value) break;
            if (timed_out && __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
                // the first wait of the call that gives up also says WHICH flag it was (word offset from the
                // time-out word, two's complement: the flags lie in front of it) -- factor_check_waits names it
                if (timed_out && atomicExch(timed_out, 1u) == 0u) timed_out[2] = (unsigned)(flag - timed_out);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// start of a kernel: every thread of the workgroup calls it
__device__ __forceinline__ void await_flag(const unsigned* flag, unsigned value, unsigned* timed_out)
{
    if (!flag) return;                              // uniform
    if (threadIdx.x == 0) spin_until(flag, value, timed_out);
    __syncthreads();
}

// end of a kernel: every thread of the workgroup calls it
__device__ __forceinline__ void signal_done(unsigned* slot, unsigned value, const unsigned* then_wait,
                                            unsigned then_value, unsigned* timed_out)
{
    if (!slot) return;                              // uniform
    // every wave's stores have left the CU (s_barrier waits for no counter), then one release for the
    // workgroup -- a release only: the consumer does its own acquire, and __threadfence() would add an L1/L2
    // invalidate (about as long again) to every kernel of the chain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (value) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (atomicAdd(slot, 1u) + 1 == total) {
            atomicExch(slot, 0u);
            if (value) __hip_atomic_store(slot + 1, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (then_wait) spin_until(then_wait, then_value, timed_out);
        }
    }
}
#endif
// factor B (slot buffers BUF_B) into L and X = L^-1 (BUF_X) for nbatch slots; prior: the matrices are prior covariances
// (cond ~ 1e8 under the reference's nugget): panel steps by substitution (gprn_ctx::acc_opt)
int factor_invert(gprn_ctx* c, int nbatch, bool prior = false);
int lauum_lower(gprn_ctx* c, int nbatch, hipStream_t stream = nullptr);   // BUF_B = lower(X^T X), X in BUF_X
int ensure_tasks(gprn_ctx* c);
// internal status: an in-kernel dependency wait gave up; the entry points of api.hip re-run the call on events
#define GPRN_E_WAIT_TIMEOUT (-100)
int factor_check_waits(gprn_ctx* c);   // GPRN_E_WAIT_TIMEOUT if an in-kernel dependency wait timed out since the last check
int factor_use_flags(gprn_ctx* c);
// smalln.hip
bool small_applies(const gprn_ctx* c);
// one half-sweep against c->d_ptrs / slot0: reads the state (mu_in, var_in), writes this phase's rows of (mu_out, var_out);
// the weight phase takes the node rows from the new state.  done: device word that makes the launch a no-op when set, or null
int small_phase(gprn_ctx* c, bool weights, const int* d_slot_gp, int nslots, const double* mu_in, const double* var_in,
                double* mu_out, double* var_out, const int* done = nullptr);
// the loop of ELBOcalc on the device (gprn_elbocalc): control words, the batch's ELBO values, the loop's last three values
struct SmallLoop { int* ctl; double *hist, *last3; int sweep, hist_at, max_iter; };
// mu^T K^-1 mu, Q1 traces, ELBO assembly of the sweep whose new state is (mu, var); loop: the stop rule too, or null
int small_tail(gprn_ctx* c, double* out4, double* scal, const double* mu, const double* var, const SmallLoop* loop = nullptr);
int small_prior(gprn_ctx* c, double** d_tab, const int* d_job_gp, double** d_kinv_out, int njobs, int* d_info);
// n_eval independent evaluations of the ELBOcalc loop side by side (gprn_elbocalc_batch); GPRN_E_UNSUPPORTED where it does not apply
int small_batch_elbocalc(gprn_ctx* c, int n_eval, const double* kparams, int n_kpar, const double* y_resid, const double* jitters,
                         const double* mu, const double* var, int max_iter, double* elbo, int* iters, int* conv, int* info,
                         double* mu_out, double* var_out);
void small_batch_free(gprn_ctx* c);
// evaluations one chunk of gprn_elbocalc_batch may hold on the small path (memory budget), >= 1
int small_batch_chunk(gprn_ctx* c);
// midn.hip: the same for problems of more than one tile, through the launch schedule with batch = evaluations x latent GPs
int mid_batch_elbocalc(gprn_ctx* c, int n_eval, const double* kparams, int n_kpar, const double* y_resid, const double* jitters,
                       const double* mu, const double* var, int max_iter, double* elbo, int* iters, int* conv, int* info,
                       double* mu_out, double* var_out);
void mid_batch_free(gprn_ctx* c);
size_t batch_budget_bytes(gprn_ctx* c);        // device memory a chunk of evaluations may take (option "batch_mem_mb")
// api_sweep.hip: one half-sweep's factorisation with its head and tail against c->d_ptrs / slot0 / d_info_cur (run_phase, midn.hip)
int phase_core(gprn_ctx* c, bool weights, const int* d_slot_gp, int ns);

// meanfield.py:640-643: np.std / np.mean of the last three values, operation by operation (one rounding each)
static inline bool elbo_stop_rule(double e0, double e1, double e2)
{
    volatile double sum = e0 + e1; sum = sum + e2;
    volatile double mean = sum / 3.0;
    volatile double d0 = e0 - mean, d1 = e1 - mean, d2 = e2 - mean;
    volatile double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
    volatile double v = q0 + q1; v = v + q2; v = v / 3.0;
    volatile double sd = __builtin_sqrt(v);
    volatile double ratio = sd / mean;
    const double crit = __builtin_fabs(ratio);
    return crit < 1e-3 && crit != 0.0;
}
