// C ABI of libgprn_hip.so (include/gprn_hip.h): context, setup, the sweep loop.
#include "gprn_internal.h"
#include "vecops.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <math.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

// ------------------------------------------------------------------ profiler
static hipEvent_t prof_event(gprn_ctx* c)
{
    if (!c->prof.pool.empty()) {
        hipEvent_t e = c->prof.pool.back();
        c->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

void prof_begin(gprn_ctx* c, int fam, hipStream_t stream)
{
    if (!c->prof.on || !((c->prof_mask >> fam) & 1)) { c->prof_open = false; return; }
    if (!stream) stream = c->stream;
    Profiler::Rec r{fam, prof_event(c), prof_event(c)};
    hipEventRecord(r.a, stream);
    c->prof.pending.push_back(r);
    c->prof_open = true;
    c->prof_stream = stream;
}

void prof_end(gprn_ctx* c)
{
    if (!c->prof_open) return;
    hipEventRecord(c->prof.pending.back().b, c->prof_stream);
    c->prof_open = false;
}

static void prof_collect(gprn_ctx* c)
{
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->stream2);
    hipStreamSynchronize(c->stream3);
    if (c->stream4) hipStreamSynchronize(c->stream4);
    for (auto& r : c->prof.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->prof.ms[r.fam] += ms;
            c->prof.n[r.fam] += 1;
        }
        c->prof.pool.push_back(r.a);
        c->prof.pool.push_back(r.b);
    }
    c->prof.pending.clear();
}

extern "C" int gprn_profile_enable(gprn_ctx* c, int mask)
{
    DeviceLock lock_(c);
    if (!c) return GPRN_E_ARG;
    prof_collect(c);
    c->prof.on = mask != 0;
    c->prof_mask = mask;
    return GPRN_OK;
}

extern "C" int gprn_profile_read(gprn_ctx* c, double* ms, int64_t* launches, int reset)
{
    DeviceLock lock_(c);
    if (!c) return GPRN_E_ARG;
    prof_collect(c);
    for (int i = 0; i < GPRN_T_COUNT; ++i) {
        if (ms) ms[i] = c->prof.ms[i];
        if (launches) launches[i] = c->prof.n[i];
        if (reset) { c->prof.ms[i] = 0; c->prof.n[i] = 0; }
    }
    return GPRN_OK;
}

// ------------------------------------------------------------------ helpers
template <typename T>
static int dev_alloc(gprn_ctx* c, T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? GPRN_E_NOMEM : GPRN_E_HIP;
    }
    return GPRN_OK;
}
#define TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

template <typename T>
static void dev_free(T*& p) { if (p) hipFree(p); p = nullptr; }

static int bad(gprn_ctx* c, const char* msg) { if (c) c->err = msg; return GPRN_E_ARG; }

// Runs `body` (an entry point that factorises); when one of its in-kernel dependency waits gave up
// (GPRN_E_WAIT_TIMEOUT: a serialising tool, a starved device, ...), the context is latched to the event
// schedule and the body runs once more -- `body` must restore what it changed before it starts over.
// On a sharded context the verdict is shared first (one max-reduce of the "timed out" word per call): the body
// issues collectives, so either every rank runs it again or none does -- a rank re-running on its own would issue
// its broadcasts and its all-reduce a second time while the others have moved on.
static int comm_allreduce(gprn_ctx* c, double* buf, size_t n, bool is_max);
static bool comm_active(const gprn_ctx* c);
// (the collective watchdog, below: EVERY stream synchronisation of this file that returns tells it the device has made
// progress -- the budget bounds a stall, not the length of a call: ADVICE r5, gprn_elbocalc at N = 16384 runs for minutes)
static inline void watch_progress(gprn_ctx* c);

static int agree_on_timeout(gprn_ctx* c, int rc, bool* any)
{
    *any = rc == GPRN_E_WAIT_TIMEOUT;
    if (!comm_active(c)) return GPRN_OK;
    if (rc == GPRN_E_COMM) return GPRN_OK;             // the transport itself is down: nothing to agree through
    if (!c->d_agree && hipMalloc(&c->d_agree, sizeof(double)) != hipSuccess) { c->err = "hipMalloc: timeout word"; return GPRN_E_NOMEM; }
    const double mine = *any ? 1.0 : 0.0;
    double all = 0.0;
    HIP_TRY(c, hipMemcpyAsync(c->d_agree, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
    int r = comm_allreduce(c, c->d_agree, 1, true);
    if (r) return r;
    HIP_TRY(c, hipMemcpyAsync(&all, c->d_agree, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    *any = all > 0.0;
    return GPRN_OK;
}

// First collective of every entry point that issues collectives: the ranks exchange the verdict of their
// LOCAL checks -- arguments, preconditions -- before anybody starts the body.  A rank that would return early on its own
// would leave the others in the body's broadcasts, which have no time-out.  Either all go on, or all return: the rank
// with the finding its own code and text, the others GPRN_E_COMM.
static int agree_to_start(gprn_ctx* c, int local_rc, const char* what)
{
    if (!comm_active(c)) return local_rc;
    if (!c->d_agree && hipMalloc(&c->d_agree, sizeof(double)) != hipSuccess) { c->err = "hipMalloc: agreement word"; return GPRN_E_NOMEM; }
    const double mine = local_rc ? 1.0 : 0.0;
    double all = 0.0;
    HIP_TRY(c, hipMemcpyAsync(c->d_agree, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int r = comm_allreduce(c, c->d_agree, 1, true);
    if (r) return local_rc ? local_rc : r;
    HIP_TRY(c, hipMemcpyAsync(&all, c->d_agree, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    if (local_rc) return local_rc;
    if (all > 0.0) {
        c->err = std::string(what) + ": another rank did not pass its checks; no rank started the call";
        return GPRN_E_COMM;
    }
    return GPRN_OK;
}

template <class F>
static int with_event_fallback(gprn_ctx* c, const char* what, F&& body, bool collective = false)
{
    int rc = body(false);
    bool again = rc == GPRN_E_WAIT_TIMEOUT;
    if (collective) {
        const int ra = agree_on_timeout(c, rc, &again);
        if (ra) return ra;
    }
    if (!again) return rc;
    hipStreamSynchronize(c->stream); hipStreamSynchronize(c->stream2); hipStreamSynchronize(c->stream3);
    if (c->stream4) hipStreamSynchronize(c->stream4);
    c->use_flags = 0;
    c->fallbacks += 1;
    if (rc == GPRN_E_WAIT_TIMEOUT)
        fprintf(stderr, "[gprn] %s: a device-side dependency wait timed out after %d ms%s; re-running the call with "
                        "HIP events (device-side waits are now off for this context)\n", what, c->wait_budget_ms,
                c->last_timeout.c_str());
    else
        fprintf(stderr, "[gprn] %s: a device-side dependency wait timed out on another rank; re-running the call with "
                        "HIP events on this rank too (device-side waits are now off for this context)\n", what);
    rc = body(true);
    if (rc == GPRN_E_WAIT_TIMEOUT) { c->err = "factorisation: dependency wait timed out on the event schedule too"; rc = GPRN_E_HIP; }
    return rc;
}

// ------------------------------------------------------------------ the collective watchdog
// A collective has no time-out of its own: a rank that dies inside one (or never enters it) leaves the others waiting --
// RCCL kernels spinning on the device, the host in the stream synchronisation behind them -- until somebody kills the job,
// holding their GPUs meanwhile.  Every entry point that issues collectives on a context with a communicator therefore
// runs under a watch: a detached thread (one per process) looks at the open watches every 200 ms, and when one has been
// open longer than the context's budget (option "comm_budget_s", default 600 s; GPRN_COMM_BUDGET_S) it says which entry
// point, which collective was enqueued last and which rank, and ends the process with a non-zero status (_exit: no
// restart, no re-exec, no unwinding through a runtime that is blocked).  The launcher then sees a failed rank and stops
// the others (bench.py's self-launcher, torchrun).
struct WatchEntry {
    std::atomic<long long> since_ms{0};        // 0: nothing open
    std::atomic<const char*> what{nullptr};    // the entry point
    std::atomic<const char*> last{nullptr};    // the collective enqueued last
    std::atomic<int> budget_s{600}, rank{0}, world{1};
};
static std::mutex g_watch_mu;
static std::vector<WatchEntry*> g_watch;
static bool g_watch_thread = false;

static long long now_ms()
{
    timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (long long)t.tv_sec * 1000 + t.tv_nsec / 1000000;
}

static void* watchdog_main(void*)
{
    for (;;) {
        usleep(200000);
        std::lock_guard<std::mutex> g(g_watch_mu);
        const long long now = now_ms();
        for (WatchEntry* w : g_watch) {
            const long long since = w->since_ms.load();
            if (!since || now - since <= 1000ll * w->budget_s.load()) continue;
            const char* what = w->what.load();
            const char* last = w->last.load();
            fprintf(stderr, "[gprn] rank %d of %d: %s has been inside a collective section for more than %d s (last collective "
                            "enqueued: %s); another rank has died or never arrived -- giving up the GPU (exit 86)\n",
                    w->rank.load(), w->world.load(), what ? what : "?", w->budget_s.load(), last ? last : "none yet");
            fflush(stderr);
            _exit(86);
        }
    }
    return nullptr;
}

static WatchEntry* watch_register(gprn_ctx* c)
{
    std::lock_guard<std::mutex> g(g_watch_mu);
    WatchEntry* w = new WatchEntry();
    const char* e = getenv("GPRN_COMM_BUDGET_S");
    w->budget_s = c->comm_budget_s > 0 ? c->comm_budget_s : (e && atoi(e) > 0 ? atoi(e) : 600);
    w->rank = c->rank; w->world = c->world;
    g_watch.push_back(w);
    if (!g_watch_thread) {
        pthread_t th;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&th, &at, watchdog_main, nullptr) == 0) g_watch_thread = true;
        pthread_attr_destroy(&at);
    }
    return w;
}

static void watch_unregister(gprn_ctx* c)
{
    if (!c->watch) return;
    std::lock_guard<std::mutex> g(g_watch_mu);
    WatchEntry* w = (WatchEntry*)c->watch;
    g_watch.erase(std::remove(g_watch.begin(), g_watch.end(), w), g_watch.end());
    delete w;
    c->watch = nullptr;
}

// held by an entry point for as long as its collectives may be outstanding (nested scopes: the outermost one counts)
struct WatchScope {
    WatchEntry* w; bool outer;
    WatchScope(gprn_ctx* c, const char* what) : w(c ? (WatchEntry*)c->watch : nullptr), outer(false)
    {
        if (!w || w->since_ms.load()) return;
        outer = true;
        w->what = what; w->last = nullptr;
        w->since_ms = now_ms();
    }
    ~WatchScope() { if (w && outer) w->since_ms = 0; }
};
static inline void watch_note(gprn_ctx* c, const char* collective)
{
    if (c->watch) ((WatchEntry*)c->watch)->last = collective;
}
// the host has just seen the device finish everything enqueued so far (a stream synchronisation returned): the open
// watch, if any, starts counting again -- the budget bounds the time WITHOUT such progress, not the length of a call
static inline void watch_progress(gprn_ctx* c)
{
    WatchEntry* w = (WatchEntry*)c->watch;
    if (w && w->since_ms.load()) w->since_ms = now_ms();
}

// Per-context switches (tests, experiments).  Returns the previous value through *old when given.
//   "flags"          1/0: device-side flags or HIP events for the factorisation's cross-stream dependencies
//   "wait_budget_ms" wall-clock budget of one in-kernel wait
//   "withhold_inner" test hook: the n-th F_INNER raise of every following call is skipped (0 = off)
//   "bulk_pad_kb" / "small_pad_kb"  unused dynamic LDS of the bulk launches (batches above / up to two matrices), KiB
//                    (-2: back to the default).  A pad that does not fit a workgroup's LDS makes the factorising calls
//                    return GPRN_E_ARG
//   "overlap"        bit mask of what runs beside the factorisations instead of before / behind them (overlap_mask below:
//                    1 B formed inside the first update, 2 row reductions panel by panel, 4 node term beside the weight
//                    phase, 8 log det B in k_finalize, 16 a sweep's end beside the next sweep); 0 = everything in sequence
//                    as in rounds 1-2; -1: the default (31).  Results are bit-identical for every value
//   "batch_mem_mb"   device memory (MiB) one chunk of gprn_elbocalc_batch's evaluations may take; longer lists run chunk by chunk
//   "comm_budget_s"  seconds an entry point may stay inside its collective section before the watchdog ends the process
//   "accurate_factor" panel steps of a factorisation by substitution instead of products with explicit inverses (diag_tile.h
//                    ACC): 0 never, 1 always (the launch path's sweeps too), -2 back to the default = every factorisation of a
//                    PRIOR matrix (the set-up, prediction, prior draws)
//   "fenced_finalize" test hook: 1 = k_reduce_finalize hands its terms over with release / acquire fences (vecops.hip)
//   "fallbacks"      read-only: calls re-run on the event schedule after a time-out
//   "batch_chunk"    read-only: evaluations per chunk in the last gprn_elbocalc_batch call
extern "C" int gprn_set_option(gprn_ctx* c, const char* name, int value, int* old)
{
    DeviceLock lock_(c);
    if (!c || !name) return GPRN_E_ARG;
    int* field = nullptr;
    if (!strcmp(name, "flags")) { factor_use_flags(c); field = &c->use_flags; }
    else if (!strcmp(name, "wait_budget_ms")) field = &c->wait_budget_ms;
    else if (!strcmp(name, "withhold_inner")) field = &c->withhold_inner;
    else if (!strcmp(name, "overlap")) field = &c->overlap_opt;
    else if (!strcmp(name, "small_path")) field = &c->small_opt;
    else if (!strcmp(name, "bulk_pad_kb")) field = &c->pad_kb_opt;
    else if (!strcmp(name, "small_pad_kb")) field = &c->pad_small_kb_opt;
    else if (!strcmp(name, "batch_mem_mb")) field = &c->batch_mem_mb;
    else if (!strcmp(name, "comm_budget_s")) field = &c->comm_budget_s;
    else if (!strcmp(name, "accurate_factor")) field = &c->acc_opt;
    else if (!strcmp(name, "fenced_finalize")) field = &c->fenced_finalize;
    else if (!strcmp(name, "fallbacks")) { if (old) *old = c->fallbacks; return GPRN_OK; }
    else if (!strcmp(name, "batch_chunk")) { if (old) *old = c->last_batch_chunk; return GPRN_OK; }
    else return bad(c, "set_option: unknown option");
    if (old) *old = *field;
    const bool is_pad = field == &c->pad_kb_opt || field == &c->pad_small_kb_opt;
    if ((is_pad || field == &c->acc_opt) && value == -2) { *field = -1; return GPRN_OK; }      // -2: back to the environment / default
    if (value >= 0) {
        if (field == &c->use_flags && value) {
            int can = 0;
            if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess) can = 0;
            if (!can) return bad(c, "set_option: this device has no stream memory operations");
        }
        if (field == &c->wait_budget_ms && value < 1) return bad(c, "set_option: wait_budget_ms >= 1");
        if (field == &c->batch_mem_mb && value < 1) return bad(c, "set_option: batch_mem_mb >= 1");
        if (field == &c->comm_budget_s && value < 1) return bad(c, "set_option: comm_budget_s >= 1");
        *field = value;
        if (field == &c->comm_budget_s && c->watch) ((WatchEntry*)c->watch)->budget_s = value;
    }
    return GPRN_OK;
}

static void free_problem(gprn_ctx* c)
{
    dev_free(c->d_time); dev_free(c->d_yraw); dev_free(c->d_yerr2); dev_free(c->d_yres);
    dev_free(c->d_variance); dev_free(c->d_mu); dev_free(c->d_var);
    dev_free(c->d_mu_save); dev_free(c->d_var_save);
    dev_free(c->d_mu_alt); dev_free(c->d_var_alt);
    if (c->d_loop_ctl) { hipFree(c->d_loop_ctl); c->d_loop_ctl = nullptr; }
    small_batch_free(c);
    mid_batch_free(c);
    if (c->h_pin_in) { hipHostFree(c->h_pin_in); c->h_pin_in = nullptr; c->pin_in_cap = 0; }
    if (c->h_pin_out) { hipHostFree(c->h_pin_out); c->h_pin_out = nullptr; c->pin_out_cap = 0; }
    dev_free(c->d_loop_hist);
    for (auto& p : c->K) dev_free(p);
    for (auto& p : c->KLinv) dev_free(p);
    for (auto& p : c->Kinv) dev_free(p);
    for (auto& p : c->Sig) dev_free(p);
    for (auto& p : c->wsB) dev_free(p);
    for (auto& p : c->wsX) dev_free(p);
    c->K.clear(); c->KLinv.clear(); c->Kinv.clear(); c->Sig.clear(); c->wsB.clear(); c->wsX.clear();
    dev_free(c->d_logdetK);
    dev_free(c->d_kinv_tab); dev_free(c->d_kinv_out); dev_free(c->d_small_ticket); dev_free(c->d_small_stamps);
    for (auto& p : c->predKs) dev_free(p);
    for (auto& p : c->predWT) dev_free(p);
    c->predKs.clear(); c->predWT.clear(); c->pred_cap = 0;
    tab_forget(c, nullptr);
    dev_free(c->tab_pred); dev_free(c->d_slotgp_all);
    dev_free(c->tab_node); dev_free(c->tab_weight); dev_free(c->tab_setup);
    dev_free(c->d_slotgp_node); dev_free(c->d_slotgp_weight); dev_free(c->d_slotgp_setup);
    dev_free(c->d_d); dev_free(c->d_s); dev_free(c->d_pred); dev_free(c->d_z); dev_free(c->d_u);
    dev_free(c->d_cs); dev_free(c->d_ct); dev_free(c->d_part);
    if (c->d_fin_terms) hipFree(c->d_fin_terms);
    if (c->d_fin_tickets) hipFree(c->d_fin_tickets);
    c->d_fin_terms = nullptr; c->d_fin_tickets = nullptr;
    dev_free(c->d_scal_base); c->d_scal = nullptr; dev_free(c->d_elbo_part); dev_free(c->d_out); dev_free(c->d_info);
    c->d_ptrs = nullptr;
    c->nslot = 0; c->out_cap = 0;
    c->factored = c->have_yres = c->have_jit = c->have_muvar = false;
    c->tables_ready = false;
    c->small_tabs_ready = c->small_sweep_ready = c->setup1_ready = false;
    dev_free(c->tab_kinv1);
}

// ------------------------------------------------------------------ context
extern "C" int gprn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ONE set of streams per device and process, shared by every context on it.  Streams are a scarce resource
// here: the runtime folds a process's streams onto a few hardware queues, and a context whose chain and side
// streams land on the same queue dead-locks the flag schedule (a kernel polling for a flag whose producer sits
// behind it in that queue) -- measured on MI355X / ROCm 7.2: the SECOND context of a process, every time, when
// each context made its own four streams.  Every entry point is synchronous (it returns results to the host), so
// contexts never have work in flight at the same time anyway; calls from several threads are serialised by the
// device's lock (DeviceLock at the top of each entry point).
//   s[0] chain (high priority): everything, incl. the latency chain of the factorisation
//   s[1] bulk (low priority): trailing updates running behind the chain (look-ahead)
//   s[2] side: in-panel work that is off the chain        s[3]: the next panel's share of an outer update
static std::mutex g_streams_mu;
static std::map<int, DeviceStreams*> g_streams;

static DeviceStreams* device_streams_acquire(int device)
{
    std::lock_guard<std::mutex> g(g_streams_mu);
    auto it = g_streams.find(device);
    if (it != g_streams.end()) { it->second->refs += 1; return it->second; }
    DeviceStreams* d = new DeviceStreams();
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    int prio[4] = {prio_hi, prio_lo, prio_hi, prio_hi};
    // (stream priorities do not order workgroup dispatch in any way a sweep can see -- six assignments measured 115.0-116.3
    // against 115.9 sweeps/s -- and CU masks that keep CUs free for the chain cost more than the free CUs give: DESIGN.md 8)
    for (int i = 0; i < 4; ++i)
        if (hipStreamCreateWithPriority(&d->s[i], hipStreamNonBlocking, prio[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) hipStreamDestroy(d->s[j]);
            delete d;
            return nullptr;
        }
    d->device = device;
    d->refs = 1;
    g_streams[device] = d;
    return d;
}

static void device_streams_release(DeviceStreams* d)
{
    if (!d) return;
    std::lock_guard<std::mutex> g(g_streams_mu);
    if (--d->refs > 0) return;
    for (int i = 0; i < 4; ++i) { hipStreamSynchronize(d->s[i]); hipStreamDestroy(d->s[i]); }
    g_streams.erase(d->device);
    delete d;
}

extern "C" int gprn_create(gprn_ctx** out, int device_id)
{
    if (!out) return GPRN_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GPRN_E_NODEV;
    if (device_id < 0 || device_id >= n) return GPRN_E_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return GPRN_E_HIP;
    gprn_ctx* c = new gprn_ctx();
    c->device = device_id;
    c->shared = device_streams_acquire(device_id);
    if (!c->shared) { delete c; return GPRN_E_HIP; }
    c->stream = c->shared->s[0];
    c->stream2 = c->shared->s[1];
    c->stream3 = c->shared->s[2];
    c->stream4 = c->shared->s[3];
    if (hipEventCreateWithFlags(&c->ev_diag, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_first, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_minil, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_inner, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_panel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_rest, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_resta, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_next, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_nodes, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_q1, hipEventDisableTiming) != hipSuccess) {
        device_streams_release(c->shared);
        delete c;
        return GPRN_E_HIP;
    }
    *out = c;
    return GPRN_OK;
}

static void comm_teardown(gprn_ctx* c);

extern "C" void gprn_destroy(gprn_ctx* c)
{
    if (!c) return;
    {
        DeviceLock lock_(c);
        hipSetDevice(c->device);
        hipStreamSynchronize(c->stream);
        hipStreamSynchronize(c->stream2);
        hipStreamSynchronize(c->stream3);
        hipStreamSynchronize(c->stream4);
        prof_collect(c);
        for (auto e : c->prof.pool) hipEventDestroy(e);
        comm_teardown(c);
        free_problem(c);
        dev_free(c->d_tasks);
        if (c->d_sig) hipFree(c->d_sig);
        if (c->d_step_stamps) hipFree(c->d_step_stamps);
        if (c->d_side_stamps) hipFree(c->d_side_stamps);
        dev_free(c->d_agree);
        dev_free(c->d_test[0]); dev_free(c->d_test[1]); dev_free(c->d_test[2]);
        hipEventDestroy(c->ev_diag);
        hipEventDestroy(c->ev_first);
        hipEventDestroy(c->ev_minil);
        hipEventDestroy(c->ev_inner);
        hipEventDestroy(c->ev_panel);
        hipEventDestroy(c->ev_rest);
        hipEventDestroy(c->ev_resta);
        hipEventDestroy(c->ev_next);
        hipEventDestroy(c->ev_nodes);
        hipEventDestroy(c->ev_q1);
        hipEventDestroy(c->ev_tail);
    }
    device_streams_release(c->shared);
    delete c;
}

extern "C" const char* gprn_last_error(const gprn_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" int gprn_last_info_gp(const gprn_ctx* c) { return c ? c->info_gp : -1; }

// ------------------------------------------------------------------ problem
extern "C" int gprn_set_data(gprn_ctx* c, int N, int p, int q, const double* time,
                             const double* y, const double* yerr)
{
    DeviceLock lock_(c);
    if (!c || N <= 0 || p <= 0 || q <= 0 || !time || !y || !yerr) return bad(c, "set_data: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    free_problem(c);
    c->N = N; c->p = p; c->q = q; c->G = q + q * p;
    c->ld = ((N + GPRN_TILE - 1) / GPRN_TILE) * GPRN_TILE;
    c->T = c->ld / GPRN_TILE;
    c->kspec.assign(c->G, KernelSpec());
    c->K.assign(c->G, nullptr);
    c->KLinv.assign(c->G, nullptr);
    c->Kinv.assign(q, nullptr);
    c->Sig.assign(c->G, nullptr);
    if (c->world == 1) c->owner.assign(c->G, 0);
    else c->owner.clear();                      // gprn_set_owners must follow
    const size_t pn = (size_t)p * N, dn = (size_t)(p + 1) * q * N;
    TRY(dev_alloc(c, &c->d_time, N));
    TRY(dev_alloc(c, &c->d_yraw, pn));
    TRY(dev_alloc(c, &c->d_yerr2, pn));
    TRY(dev_alloc(c, &c->d_yres, pn));
    TRY(dev_alloc(c, &c->d_variance, pn));
    TRY(dev_alloc(c, &c->d_mu, dn));
    TRY(dev_alloc(c, &c->d_var, dn));
    TRY(dev_alloc(c, &c->d_mu_save, dn));
    TRY(dev_alloc(c, &c->d_var_save, dn));
    TRY(dev_alloc(c, &c->d_mu_alt, dn));
    TRY(dev_alloc(c, &c->d_var_alt, dn));
    TRY(dev_alloc(c, &c->d_logdetK, c->G));
    TRY(dev_alloc(c, &c->d_scal_base, 2 * (size_t)(3 * c->G + q * q)));
    HIP_TRY(c, hipMemset(c->d_scal_base, 0, 2 * (size_t)(3 * c->G + q * q) * sizeof(double)));
    TRY(dev_alloc(c, &c->d_elbo_part, 2 * (size_t)GPRN_ELBO_PART_DOUBLES));
    c->d_scal = c->d_scal_base;
    c->d_logdetB = c->d_scal;
    c->d_trBinv = c->d_scal + c->G;
    c->d_muKmu = c->d_scal + 2 * c->G;
    c->d_q1 = c->d_scal + 3 * c->G;
    std::vector<double> e2(pn);
    for (size_t i = 0; i < pn; ++i) e2[i] = yerr[i] * yerr[i];
    HIP_TRY(c, hipMemcpy(c->d_time, time, N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_yraw, y, pn * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_yerr2, e2.data(), pn * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_logdetK, 0, c->G * sizeof(double)));
    c->h_yerr2 = e2;
    return GPRN_OK;
}

extern "C" int gprn_set_kernel(gprn_ctx* c, int gp, const int32_t* ops, int n_ops,
                               const double* params, int n_params, int add_nugget)
{
    DeviceLock lock_(c);
    if (!c || !c->N) return bad(c, "set_kernel: call set_data first");
    if (gp < 0 || gp >= c->G || !ops || n_ops <= 0 || n_ops > GPRN_MAX_OPS || n_params < 0 ||
        n_params > GPRN_MAX_KPARAMS || (n_params && !params))
        return bad(c, "set_kernel: bad argument");
    int depth = 0;
    for (int o = 0; o < n_ops; ++o) {
        const int op = ops[3 * o], kid = ops[3 * o + 1], off = ops[3 * o + 2];
        if (op == GPRN_OP_PUSH) {
            if (kid < 0 || kid >= GPRN_K_COUNT || off < 0 || off > n_params) return bad(c, "set_kernel: bad push");
            if (++depth > 8) return bad(c, "set_kernel: expression too deep");
        } else if (op == GPRN_OP_ADD || op == GPRN_OP_MUL) {
            if (--depth < 1) return bad(c, "set_kernel: malformed expression");
        } else return bad(c, "set_kernel: unknown opcode");
    }
    if (depth != 1) return bad(c, "set_kernel: malformed expression");
    KernelSpec& ks = c->kspec[gp];
    if (ks.set && !ks.uploaded && ks.n_ops == n_ops && ks.n_params == n_params &&
        ks.nugget == (add_nugget ? 1 : 0) && !memcmp(ks.ops, ops, 3 * n_ops * sizeof(int32_t)) &&
        (!n_params || !memcmp(ks.params, params, n_params * sizeof(double))))
        return GPRN_OK;                      // unchanged: a finished factorisation stays valid
    ks.set = true; ks.uploaded = false;
    ks.n_ops = n_ops; ks.n_params = n_params; ks.nugget = add_nugget ? 1 : 0;
    memcpy(ks.ops, ops, 3 * n_ops * sizeof(int32_t));
    if (n_params) memcpy(ks.params, params, n_params * sizeof(double));
    c->factored = false;
    return GPRN_OK;
}

static int ensure_gp_storage(gprn_ctx* c, int g)
{
    const size_t nn = (size_t)c->ld * c->ld;
    if (!c->K[g]) TRY(dev_alloc(c, &c->K[g], nn));
    if (!c->KLinv[g]) TRY(dev_alloc(c, &c->KLinv[g], nn));
    return GPRN_OK;
}

extern "C" int gprn_upload_K(gprn_ctx* c, int gp, const double* Kh)
{
    DeviceLock lock_(c);
    if (!c || !c->N) return bad(c, "upload_K: call set_data first");
    if (gp < 0 || gp >= c->G || !Kh) return bad(c, "upload_K: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    KernelSpec& ks = c->kspec[gp];
    ks.set = true; ks.uploaded = true;
    c->factored = false;
    if (c->owner.empty()) return bad(c, "upload_K: call set_owners first");
    if (c->owner[gp] != c->rank) return GPRN_OK;                    // not needed on this rank
    TRY(ensure_gp_storage(c, gp));
    // identity padding, then the N x N block
    const int N = c->N, ld = c->ld;
    std::vector<double> pad((size_t)ld * ld, 0.0);
    for (int m = 0; m < ld; ++m) {
        if (m < N) memcpy(&pad[(size_t)m * ld], Kh + (size_t)m * N, N * sizeof(double));
        else pad[(size_t)m * ld + m] = 1.0;
    }
    HIP_TRY(c, hipMemcpy(c->K[gp], pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
    return GPRN_OK;
}

extern "C" int gprn_set_y_resid(gprn_ctx* c, const double* y)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !y) return bad(c, "set_y_resid: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy(c->d_yres, y, (size_t)c->p * c->N * sizeof(double), hipMemcpyHostToDevice));
    c->have_yres = true;
    return GPRN_OK;
}

extern "C" int gprn_set_jitters(gprn_ctx* c, const double* jit)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !jit) return bad(c, "set_jitters: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    std::vector<double> v((size_t)c->p * c->N);
    for (int i = 0; i < c->p; ++i)
        for (int n = 0; n < c->N; ++n)
            v[(size_t)i * c->N + n] = jit[i] * jit[i] + c->h_yerr2[(size_t)i * c->N + n];
    HIP_TRY(c, hipMemcpy(c->d_variance, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    c->have_jit = true;
    return GPRN_OK;
}

extern "C" int gprn_set_muvar(gprn_ctx* c, const double* mu, const double* var)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !mu || !var) return bad(c, "set_muvar: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    const size_t dn = (size_t)(c->p + 1) * c->q * c->N * sizeof(double);
    HIP_TRY(c, hipMemcpy(c->d_mu, mu, dn, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_var, var, dn, hipMemcpyHostToDevice));
    c->have_muvar = true;
    return GPRN_OK;
}

extern "C" int gprn_get_muvar(gprn_ctx* c, double* mu, double* var)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !mu || !var) return bad(c, "get_muvar: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    const size_t dn = (size_t)(c->p + 1) * c->q * c->N * sizeof(double);
    HIP_TRY(c, hipMemcpy(mu, c->d_mu, dn, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(var, c->d_var, dn, hipMemcpyDeviceToHost));
    return GPRN_OK;
}

// ------------------------------------------------------------------ sharding / RCCL
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl;
static void* g_rccl_handle = nullptr;

static int rccl_load(std::string* err)
{
    if (g_rccl_handle) return GPRN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl_handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl_handle) break;
    }
    if (!g_rccl_handle) { if (err) *err = std::string("dlopen librccl: ") + dlerror(); return GPRN_E_COMM; }
#define SYM(field, name)                                                             \
    *(void**)(&g_rccl.field) = dlsym(g_rccl_handle, name);                           \
    if (!g_rccl.field) { if (err) *err = std::string("dlsym ") + name; return GPRN_E_COMM; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return GPRN_OK;
}

#define NCCL_TRY(c, expr)                                                            \
    do {                                                                             \
        ncclResult_t r_ = (expr);                                                    \
        if (r_ != ncclSuccess) {                                                     \
            (c)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r_);        \
            return GPRN_E_COMM;                                                      \
        }                                                                            \
    } while (0)

// ---- rehearsal transport (GPRN_COMM_TRANSPORT=shm) ----
// The same collectives through a POSIX shared-memory segment and host copies, so that several
// ranks can run the sharded path on ONE GPU (RCCL refuses two ranks on one device).  For the
// tests of a one-GPU box only: every operation synchronises the stream and crosses PCIe twice.
struct ShmComm {
    struct Header { std::atomic<int> count; std::atomic<int> sense; };
    static constexpr size_t kHeader = 4096, kSlot = 1 << 20;   // bytes; one slot per rank
    int world = 1, rank = 0, local_sense = 0;
    Header* hdr = nullptr;
    char* slots = nullptr;
    std::string name;
    size_t bytes() const { return kHeader + (size_t)world * kSlot; }
};

static bool shm_transport()
{
    const char* e = getenv("GPRN_COMM_TRANSPORT");
    return e && !strcmp(e, "shm");
}

static int shm_barrier(gprn_ctx* c, ShmComm* sc)
{
    sc->local_sense ^= 1;
    if (sc->hdr->count.fetch_add(1) + 1 == sc->world) {
        sc->hdr->count.store(0);
        sc->hdr->sense.store(sc->local_sense);
        watch_progress(c);
        return GPRN_OK;
    }
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0; sc->hdr->sense.load() != sc->local_sense; ++spin) {
        sched_yield();
        if ((spin & 1023) == 1023) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            const int budget = c->watch ? ((WatchEntry*)c->watch)->budget_s.load() : 120;
            if (t1.tv_sec - t0.tv_sec > budget + 5) { c->err = "shm transport: barrier timed out (a rank died?)"; return GPRN_E_COMM; }
        }
    }
    watch_progress(c);
    return GPRN_OK;
}

static int shm_open_comm(gprn_ctx* c, int world, int rank, const char* id128)
{
    ShmComm* sc = new ShmComm();
    sc->world = world; sc->rank = rank;
    char nm[64] = "/gprn_";
    for (int i = 0; i < 16; ++i) snprintf(nm + 6 + 2 * i, 3, "%02x", (unsigned char)id128[8 + i]);
    sc->name = nm;
    int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sc->bytes()) != 0) {
        if (fd >= 0) close(fd);
        delete sc; c->err = "shm transport: cannot create the segment"; return GPRN_E_COMM;
    }
    void* m = mmap(nullptr, sc->bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete sc; c->err = "shm transport: mmap failed"; return GPRN_E_COMM; }
    sc->hdr = (ShmComm::Header*)m;                 // a fresh segment is zero-filled: count 0, sense 0
    sc->slots = (char*)m + ShmComm::kHeader;
    c->shm = sc;
    int rc = shm_barrier(c, sc);                   // everybody is attached
    if (rc == GPRN_OK && rank == 0) shm_unlink(nm);    // the mapping outlives the name
    return rc;
}

static void shm_close_comm(gprn_ctx* c)
{
    ShmComm* sc = (ShmComm*)c->shm;
    if (!sc) return;
    munmap((void*)sc->hdr, sc->bytes());
    delete sc;
    c->shm = nullptr;
}

// buf (device, n doubles) of `root` -> buf of every rank
static int shm_broadcast(gprn_ctx* c, double* buf, size_t n, int root)
{
    ShmComm* sc = (ShmComm*)c->shm;
    if (n * sizeof(double) > ShmComm::kSlot) { c->err = "shm transport: message too large"; return GPRN_E_COMM; }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    if (sc->rank == root) HIP_TRY(c, hipMemcpy(sc->slots, buf, n * sizeof(double), hipMemcpyDeviceToHost));
    TRY(shm_barrier(c, sc));
    if (sc->rank != root) HIP_TRY(c, hipMemcpy(buf, sc->slots, n * sizeof(double), hipMemcpyHostToDevice));
    return shm_barrier(c, sc);
}

// sum or max over ranks, in rank order on every rank (identical bits everywhere)
static int shm_allreduce(gprn_ctx* c, double* buf, size_t n, bool is_max)
{
    ShmComm* sc = (ShmComm*)c->shm;
    if (n * sizeof(double) > ShmComm::kSlot) { c->err = "shm transport: message too large"; return GPRN_E_COMM; }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy(sc->slots + (size_t)sc->rank * ShmComm::kSlot, buf, n * sizeof(double), hipMemcpyDeviceToHost));
    TRY(shm_barrier(c, sc));
    std::vector<double> acc(n);
    for (size_t i = 0; i < n; ++i) {
        double v = ((const double*)sc->slots)[i];
        for (int r = 1; r < sc->world; ++r) {
            const double w = ((const double*)(sc->slots + (size_t)r * ShmComm::kSlot))[i];
            v = is_max ? std::max(v, w) : v + w;
        }
        acc[i] = v;
    }
    HIP_TRY(c, hipMemcpy(buf, acc.data(), n * sizeof(double), hipMemcpyHostToDevice));
    return shm_barrier(c, sc);
}

// ---- the three collectives of the path, on whichever transport the context has ----
static int comm_broadcast(gprn_ctx* c, double* buf, size_t n, int root)
{
    watch_note(c, "row broadcast");
    if (c->shm) return shm_broadcast(c, buf, n, root);
    NCCL_TRY(c, g_rccl.Broadcast(buf, buf, n, ncclDouble, root, (ncclComm_t)c->comm, c->stream));
    return GPRN_OK;
}

static int comm_allreduce(gprn_ctx* c, double* buf, size_t n, bool is_max)
{
    watch_note(c, is_max ? "max all-reduce (agreement / barrier)" : "sum all-reduce (per-GP scalars)");
    if (c->shm) return shm_allreduce(c, buf, n, is_max);
    NCCL_TRY(c, g_rccl.AllReduce(buf, buf, n, ncclDouble, is_max ? ncclMax : ncclSum, (ncclComm_t)c->comm, c->stream));
    return GPRN_OK;
}

static bool comm_active(const gprn_ctx* c) { return c->comm || c->shm; }

extern "C" int gprn_comm_unique_id(char* id128)
{
    if (!id128) return GPRN_E_ARG;
    if (shm_transport()) {                         // 128 random bytes name the segment
        memset(id128, 0, 128);
        memcpy(id128, "gprnshm", 8);
        int fd = open("/dev/urandom", O_RDONLY);
        if (fd < 0 || read(fd, id128 + 8, 16) != 16) { if (fd >= 0) close(fd); return GPRN_E_COMM; }
        close(fd);
        return GPRN_OK;
    }
    if (rccl_load(nullptr)) return GPRN_E_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return GPRN_E_COMM;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return GPRN_OK;
}

static void comm_teardown(gprn_ctx* c)
{
    watch_unregister(c);
    if (c->comm && g_rccl_handle) g_rccl.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    shm_close_comm(c);
}

extern "C" int gprn_comm_init(gprn_ctx* c, int world, int rank, const char* id128)
{
    DeviceLock lock_(c);
    if (!c || world < 1 || rank < 0 || rank >= world) return bad(c, "comm_init: bad argument");
    if (c->N) return bad(c, "comm_init: call before set_data");
    HIP_TRY(c, hipSetDevice(c->device));
    comm_teardown(c);
    c->world = world; c->rank = rank;
    // a one-rank communicator is legal RCCL and lets a single GPU exercise every collective call
    if (world == 1 && !getenv("GPRN_FORCE_RCCL")) return GPRN_OK;
    if (!id128) return bad(c, "comm_init: id required");
    if (!memcmp(id128, "gprnshm", 8)) {
        const int rc = shm_open_comm(c, world, rank, id128);
        if (!rc) c->watch = watch_register(c);
        return rc;
    }
    TRY(rccl_load(&c->err));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    NCCL_TRY(c, g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->watch = watch_register(c);
    return GPRN_OK;
}

extern "C" int gprn_set_owners(gprn_ctx* c, const int* owner)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !owner) return bad(c, "set_owners: call set_data first");
    for (int g = 0; g < c->G; ++g)
        if (owner[g] < 0 || owner[g] >= c->world) return bad(c, "set_owners: rank out of range");
    c->owner.assign(owner, owner + c->G);
    c->factored = false;
    c->tables_ready = false;
    return GPRN_OK;
}

extern "C" int gprn_comm_barrier_max(gprn_ctx* c, double* value)
{
    DeviceLock lock_(c);
    WatchScope watch_(c, "gprn_comm_barrier_max");
    if (!c || !value) return GPRN_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (comm_active(c)) {
        double* d = nullptr;
        TRY(dev_alloc(c, &d, 1));
        HIP_TRY(c, hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
        TRY(comm_allreduce(c, d, 1, true));
        HIP_TRY(c, hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
        hipFree(d);
    } else {
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    }
    return GPRN_OK;
}

// sum of a caller-owned host vector over the ranks (every rank gets the result): what a pool
// of independent ELBO evaluations (emcee walkers) needs to share its values
extern "C" int gprn_comm_allreduce_sum(gprn_ctx* c, double* buf, int n)
{
    DeviceLock lock_(c);
    WatchScope watch_(c, "gprn_comm_allreduce_sum");
    if (!c || !buf || n < 0) return bad(c, "comm_allreduce_sum: bad argument");
    if (!comm_active(c) || n == 0) return GPRN_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    double* d = nullptr;
    TRY(dev_alloc(c, &d, (size_t)n));
    int rc = GPRN_OK;
    if (hipMemcpyAsync(d, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = GPRN_E_HIP;
    if (rc == GPRN_OK) rc = comm_allreduce(c, d, (size_t)n, false);
    if (rc == GPRN_OK && hipMemcpyAsync(buf, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = GPRN_E_HIP;
    if (rc == GPRN_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = GPRN_E_HIP;
    hipFree(d);
    if (rc == GPRN_E_HIP) c->err = "comm_allreduce_sum: copy failed";
    return rc;
}

// rows of the (p+1, q, N) state owned by other ranks arrive from their owners
static int exchange_rows(gprn_ctx* c, bool weights)
{
    if (!comm_active(c)) return GPRN_OK;
    const int g0 = weights ? c->q : 0, g1 = weights ? c->G : c->q;
    if (c->comm) NCCL_TRY(c, g_rccl.GroupStart());
    for (int g = g0; g < g1; ++g) {
        size_t row;
        if (g < c->q) row = g;
        else { const int kk = g - c->q, j = kk / c->p, i = kk % c->p; row = (size_t)(1 + i) * c->q + j; }
        double* m = c->d_mu + row * c->N;
        double* v = c->d_var + row * c->N;
        TRY(comm_broadcast(c, m, c->N, c->owner[g]));
        TRY(comm_broadcast(c, v, c->N, c->owner[g]));
    }
    if (c->comm) NCCL_TRY(c, g_rccl.GroupEnd());
    return GPRN_OK;
}

static int reduce_scalars(gprn_ctx* c)
{
    if (!comm_active(c)) return GPRN_OK;
    const size_t n = 3 * (size_t)c->G + (size_t)c->q * c->q;
    return comm_allreduce(c, c->d_scal, n, false);
}

// ------------------------------------------------------------------ tables
static int upload_table(gprn_ctx* c, double** d_tab, const std::vector<double*>& rows)
{
    HIP_TRY(c, hipMemcpyAsync(d_tab, rows.data(), rows.size() * sizeof(double*),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    tab_note(c, d_tab, rows.data(), rows.size());
    return GPRN_OK;
}

static int build_tables(gprn_ctx* c)
{
    if (c->tables_ready) return GPRN_OK;
    // (the second set of node workspaces and its table are rebuilt on demand: sweep_impl)
    c->loc_nodes.clear(); c->loc_weights.clear();
    for (int g = 0; g < c->q; ++g) if (c->owner[g] == c->rank) c->loc_nodes.push_back(g);
    for (int g = c->q; g < c->G; ++g) if (c->owner[g] == c->rank) c->loc_weights.push_back(g);
    const int want = std::max<size_t>(1, c->loc_nodes.size() + c->loc_weights.size());
    const size_t nn = (size_t)c->ld * c->ld;
    if (want != c->nslot) {
        for (auto& p : c->wsB) dev_free(p);
        for (auto& p : c->wsX) dev_free(p);
        c->wsB.assign(want, nullptr); c->wsX.assign(want, nullptr);
        for (int s = 0; s < want; ++s) {
            TRY(dev_alloc(c, &c->wsB[s], nn));
            TRY(dev_alloc(c, &c->wsX[s], nn));
        }
        tab_forget(c, c->tab_node); tab_forget(c, c->tab_weight); tab_forget(c, c->tab_setup);
        dev_free(c->tab_node); dev_free(c->tab_weight); dev_free(c->tab_setup);
        dev_free(c->d_slotgp_node); dev_free(c->d_slotgp_weight); dev_free(c->d_slotgp_setup);
        dev_free(c->d_d); dev_free(c->d_s); dev_free(c->d_pred); dev_free(c->d_z); dev_free(c->d_u);
        dev_free(c->d_cs); dev_free(c->d_ct); dev_free(c->d_part); dev_free(c->d_info);
        if (c->d_fin_terms) hipFree(c->d_fin_terms);
        if (c->d_fin_tickets) hipFree(c->d_fin_tickets);
        c->d_fin_terms = nullptr; c->d_fin_tickets = nullptr;
        const size_t tab = (size_t)want * GPRN_NBUF, vec = (size_t)want * c->ld;
        TRY(dev_alloc(c, &c->tab_node, tab)); TRY(dev_alloc(c, &c->tab_weight, tab));
        TRY(dev_alloc(c, &c->tab_setup, tab));
        TRY(dev_alloc(c, &c->d_slotgp_node, want)); TRY(dev_alloc(c, &c->d_slotgp_weight, want));
        TRY(dev_alloc(c, &c->d_slotgp_setup, want));
        TRY(dev_alloc(c, &c->d_d, vec)); TRY(dev_alloc(c, &c->d_s, vec)); TRY(dev_alloc(c, &c->d_pred, vec));
        TRY(dev_alloc(c, &c->d_z, vec)); TRY(dev_alloc(c, &c->d_u, vec)); TRY(dev_alloc(c, &c->d_cs, vec));
        TRY(dev_alloc(c, &c->d_ct, vec));
        TRY(dev_alloc(c, &c->d_part, (size_t)want * c->T * 2 * c->ld));
        TRY(dev_alloc(c, &c->d_info, 3 * (size_t)want));
        HIP_TRY(c, hipMemset(c->d_info, 0, 3 * (size_t)want * sizeof(int)));
        c->nslot = want;
    }
    for (int g : c->loc_nodes) TRY(ensure_gp_storage(c, g));
    for (int g : c->loc_weights) TRY(ensure_gp_storage(c, g));
    std::vector<double*> rows((size_t)c->nslot * GPRN_NBUF, nullptr);
    // nodes keep slots [0, #nodes), weights the slots after them: both phases' factors stay
    // resident, so the node phase's X^T X (quirk Q1) can run behind the weight phase
    auto fill_rows = [&](const std::vector<int>& gps, size_t first) {
        std::fill(rows.begin(), rows.end(), nullptr);
        for (size_t s = 0; s < gps.size(); ++s) {
            rows[s * GPRN_NBUF + BUF_B] = c->wsB[first + s];
            rows[s * GPRN_NBUF + BUF_X] = c->wsX[first + s];
            rows[s * GPRN_NBUF + BUF_K] = c->K[gps[s]];
            rows[s * GPRN_NBUF + BUF_KLINV] = c->KLinv[gps[s]];
        }
    };
    fill_rows(c->loc_nodes, 0);
    TRY(upload_table(c, c->tab_node, rows));
    fill_rows(c->loc_weights, c->loc_nodes.size());
    TRY(upload_table(c, c->tab_weight, rows));
    if (!c->loc_nodes.empty())
        HIP_TRY(c, hipMemcpy(c->d_slotgp_node, c->loc_nodes.data(), c->loc_nodes.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!c->loc_weights.empty())
        HIP_TRY(c, hipMemcpy(c->d_slotgp_weight, c->loc_weights.data(), c->loc_weights.size() * sizeof(int), hipMemcpyHostToDevice));
    c->tables_ready = true;
    c->small_tabs_ready = false;
    c->small_sweep_ready = false;
    c->setup1_ready = false;
    return GPRN_OK;
}

static int check_info(gprn_ctx* c, const int* d_info, const std::vector<int>& gps, int* first)
{
    std::vector<int> h(gps.size());
    if (gps.empty()) return GPRN_OK;
    HIP_TRY(c, hipMemcpy(h.data(), d_info, gps.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < gps.size(); ++s)
        if (h[s] > 0 && *first == 0) { *first = h[s]; c->info_gp = gps[s]; }
    return GPRN_OK;
}

// ------------------------------------------------------------------ setup
// fill + chol(K) + chol(K)^-1 (+ K^-1 for the nodes that feed quirk Q1)
static int factor_priors_impl(gprn_ctx* c);

extern "C" int gprn_factor_priors(gprn_ctx* c)
{
    DeviceLock lock_(c);
    WatchScope watch_(c, "gprn_factor_priors");
    if (!c || !c->N) return bad(c, "factor_priors: call set_data first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->owner.empty()) return bad(c, "factor_priors: call set_owners first");
    int pre = GPRN_OK;
    for (int g = 0; g < c->G && !pre; ++g)
        if (!c->kspec[g].set) pre = bad(c, "factor_priors: a latent GP has no kernel");
    if ((pre = agree_to_start(c, pre, "factor_priors"))) return pre;
    // (every K is refilled from its kernel spec -- or still holds the uploaded matrix -- so a re-run starts clean)
    return with_event_fallback(c, "factor_priors", [&](bool) { return factor_priors_impl(c); }, true);
}

// What a SWEEP of the small path reads beside the phase tables: the ticket of k_small_tail and the table of K_j^-1 pointers
// (quirk Q1).  Whichever set-up ran last -- the small one below or the launch schedule's (option "small_path" = 0 or
// gprn_keep_sigma at that time; it fills Kinv[j], j >= 1, too) -- the sweep may take either path afterwards (a
// set-up through the launch path followed by a sweep on the small path read a null ticket and a null table).
static int ensure_small_sweep_tabs(gprn_ctx* c)
{
    if (c->small_sweep_ready) return GPRN_OK;
    std::vector<double*> ktab(c->q, nullptr);
    for (int j = 1; j < c->q; ++j) {
        if (!c->Kinv[j]) return bad(c, "small path: K_j^-1 of a node is missing (no set-up yet?)");
        ktab[j] = c->Kinv[j];
    }
    dev_free(c->d_kinv_tab);
    TRY(dev_alloc(c, &c->d_kinv_tab, (size_t)c->q));
    HIP_TRY(c, hipMemcpy(c->d_kinv_tab, ktab.data(), ktab.size() * sizeof(double*), hipMemcpyHostToDevice));
    if (!c->d_small_ticket) {
        TRY(dev_alloc(c, &c->d_small_ticket, 1));
        HIP_TRY(c, hipMemset(c->d_small_ticket, 0, sizeof(unsigned)));
    }
    c->small_sweep_ready = true;
    return GPRN_OK;
}

// The set-up of a problem of one or two tiles on one rank (smalln.hip): the fills, then ONE launch -- a workgroup per latent
// GP copies K, factors and inverts it, takes log det K and, where quirk Q1 needs it, forms K_j^-1 -- and one read-back.
static int factor_priors_small(gprn_ctx* c, bool sync = true)
{
    TRY(build_tables(c));
    c->info_gp = -1;
    const size_t nn = (size_t)c->ld * c->ld;
    std::vector<int> gps(c->loc_nodes);
    gps.insert(gps.end(), c->loc_weights.begin(), c->loc_weights.end());
    const int nj = (int)gps.size();
    if (!c->small_tabs_ready) {
        std::vector<double*> rows((size_t)c->nslot * GPRN_NBUF, nullptr), kout(c->nslot, nullptr);
        for (int s = 0; s < nj; ++s) {
            const int g = gps[s];
            rows[s * GPRN_NBUF + BUF_B] = c->wsB[s];
            rows[s * GPRN_NBUF + BUF_X] = c->KLinv[g];
            rows[s * GPRN_NBUF + BUF_K] = c->K[g];
            rows[s * GPRN_NBUF + BUF_KLINV] = c->KLinv[g];
            if (g >= 1 && g < c->q) {                  // quirk Q1: node k < j needs K_j^-1
                if (!c->Kinv[g]) { TRY(dev_alloc(c, &c->Kinv[g], nn)); c->small_sweep_ready = false; }
                kout[s] = c->Kinv[g];
            }
        }
        TRY(upload_table(c, c->tab_setup, rows));
        c->setup1_ready = false;                       // (the launch-path set-up's rows are gone)
        HIP_TRY(c, hipMemcpy(c->d_slotgp_setup, gps.data(), nj * sizeof(int), hipMemcpyHostToDevice));
        dev_free(c->d_kinv_out);
        TRY(dev_alloc(c, &c->d_kinv_out, (size_t)c->nslot));
        HIP_TRY(c, hipMemcpy(c->d_kinv_out, kout.data(), kout.size() * sizeof(double*), hipMemcpyHostToDevice));
        c->small_tabs_ready = true;
    }
    TRY(ensure_small_sweep_tabs(c));
    for (int g : gps)
        if (!c->kspec[g].uploaded) TRY(launch_fill(c, c->kspec[g], c->K[g]));
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, (size_t)c->nslot * sizeof(int), c->stream));    // (the kernels only raise the verdicts)
    TRY(small_prior(c, c->tab_setup, c->d_slotgp_setup, c->d_kinv_out, nj, c->d_info));
    c->factored = true;
    if (!sync) return GPRN_OK;                             // gprn_elbocalc reads the pivot verdicts with its own results
    int first_info = 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);          // (the library's streams do not synchronise with the null stream's copies)
    TRY(check_info(c, c->d_info, gps, &first_info));
    return first_info;
}

// The set-up of an UNSHARDED problem through the launch schedule: every latent GP is local, so the tables of the call never
// change (uploaded once per problem: setup1_ready), all K_j^-1 of quirk Q1 are ONE X^T X launch over the nodes j >= 1, and the
// host waits once, for the pivot verdicts.  (The general form below synchronises a dozen times per call -- table uploads,
// one X^T X per node with its own table, the host's filter of log det K for the all-reduce: 0.25 of the 0.4-0.57 ms a set-up
// took at N = 200 ... 512, where an evaluation of nELBO is 1.3-2.4 ms.)
static int factor_priors_single(gprn_ctx* c)
{
    TRY(build_tables(c));
    TRY(ensure_tasks(c));
    c->small_tabs_ready = false;
    c->small_sweep_ready = false;
    c->info_gp = -1;
    const size_t nn = (size_t)c->ld * c->ld;
    std::vector<int> gps(c->loc_nodes);
    gps.insert(gps.end(), c->loc_weights.begin(), c->loc_weights.end());
    const int nb = (int)gps.size(), n_inv = c->q - 1;
    for (int j = 1; j < c->q; ++j)
        if (!c->Kinv[j]) { TRY(dev_alloc(c, &c->Kinv[j], nn)); c->setup1_ready = false; }
    if (!c->setup1_ready) {
        std::vector<double*> rows((size_t)c->nslot * GPRN_NBUF, nullptr);
        for (int s = 0; s < nb; ++s) {
            rows[s * GPRN_NBUF + BUF_B] = c->wsB[s];
            rows[s * GPRN_NBUF + BUF_X] = c->KLinv[gps[s]];
            rows[s * GPRN_NBUF + BUF_K] = c->K[gps[s]];
            rows[s * GPRN_NBUF + BUF_KLINV] = c->KLinv[gps[s]];
        }
        TRY(upload_table(c, c->tab_setup, rows));
        HIP_TRY(c, hipMemcpy(c->d_slotgp_setup, gps.data(), nb * sizeof(int), hipMemcpyHostToDevice));
        if (n_inv > 0) {                                   // lower(K_j^-1) = lower(X^T X), X = chol(K_j)^-1: nodes 1 .. q - 1
            dev_free(c->tab_kinv1);
            TRY(dev_alloc(c, &c->tab_kinv1, (size_t)n_inv * GPRN_NBUF));
            std::vector<double*> kr((size_t)n_inv * GPRN_NBUF, nullptr);
            for (int j = 1; j < c->q; ++j) {
                kr[(size_t)(j - 1) * GPRN_NBUF + BUF_B] = c->Kinv[j];
                kr[(size_t)(j - 1) * GPRN_NBUF + BUF_X] = c->KLinv[j];
            }
            HIP_TRY(c, hipMemcpy(c->tab_kinv1, kr.data(), kr.size() * sizeof(double*), hipMemcpyHostToDevice));
        }
        c->setup1_ready = true;
    }
    for (int s = 0; s < nb; ++s) {
        const int g = gps[s];
        if (!c->kspec[g].uploaded) TRY(launch_fill(c, c->kspec[g], c->K[g]));
        HIP_TRY(c, hipMemcpyAsync(c->wsB[s], c->K[g], nn * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, 3 * (size_t)c->nslot * sizeof(int), c->stream));
    c->d_ptrs = c->tab_setup;
    c->slot0 = 0;
    c->d_info_cur = c->d_info;
    TRY(factor_invert(c, nb, true));
    TRY(vec_logdet(c, BUF_B, c->d_slotgp_setup, nb, c->d_logdetK));
    if (n_inv > 0) {
        c->d_ptrs = c->tab_kinv1;
        TRY(lauum_lower(c, n_inv));
        c->d_ptrs = c->tab_setup;
    }
    int first_info = 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);          // (the one wait of the call; the null stream's copy below does not
    TRY(check_info(c, c->d_info, gps, &first_info));      // wait for the library's non-blocking streams by itself)
    TRY(factor_check_waits(c));
    c->factored = true;
    return first_info;
}

static int factor_priors_impl(gprn_ctx* c)
{
    if (small_applies(c) && c->world == 1) return factor_priors_small(c);
    if (!comm_active(c) && c->world == 1) return factor_priors_single(c);
    TRY(build_tables(c));
    TRY(ensure_tasks(c));
    c->small_tabs_ready = false;               // (tab_setup gets this path's rows; Kinv[j] may be allocated below)
    c->small_sweep_ready = false;
    c->setup1_ready = false;
    c->info_gp = -1;
    const size_t nn = (size_t)c->ld * c->ld;
    HIP_TRY(c, hipMemsetAsync(c->d_logdetK, 0, c->G * sizeof(double), c->stream));

    // which nodes need an explicit K_j^-1 here: j >= 1 with a local node k < j
    std::vector<char> need_inv(c->q, 0);
    if (c->q > 1 && !c->loc_nodes.empty())
        for (int j = c->loc_nodes.front() + 1; j < c->q; ++j) need_inv[j] = 1;

    struct Job { int g; bool owned; };
    std::vector<Job> jobs;
    for (int g : c->loc_nodes) jobs.push_back({g, true});
    for (int g : c->loc_weights) jobs.push_back({g, true});
    for (int j = 0; j < c->q; ++j)
        if (need_inv[j] && c->owner[j] != c->rank) jobs.push_back({j, false});

    int first_info = 0;
    for (size_t j0 = 0; j0 < jobs.size(); j0 += c->nslot) {
        const int nb = (int)std::min<size_t>(c->nslot, jobs.size() - j0);
        std::vector<double*> rows((size_t)c->nslot * GPRN_NBUF, nullptr);
        std::vector<int> gps(nb);
        for (int s = 0; s < nb; ++s) {
            const Job& jb = jobs[j0 + s];
            gps[s] = jb.g;
            double* Kdst = jb.owned ? c->K[jb.g] : c->wsB[s];
            if (jb.owned && c->kspec[jb.g].uploaded) {
                // already on the device
            } else if (c->kspec[jb.g].uploaded) {
                return bad(c, "factor_priors: a host-evaluated node kernel cannot feed another rank (q > 1, sharded)");
            } else {
                TRY(launch_fill(c, c->kspec[jb.g], Kdst));
            }
            if (jb.owned)
                HIP_TRY(c, hipMemcpyAsync(c->wsB[s], c->K[jb.g], nn * sizeof(double),
                                          hipMemcpyDeviceToDevice, c->stream));
            rows[s * GPRN_NBUF + BUF_B] = c->wsB[s];
            rows[s * GPRN_NBUF + BUF_X] = jb.owned ? c->KLinv[jb.g] : c->wsX[s];
            rows[s * GPRN_NBUF + BUF_K] = Kdst;
            rows[s * GPRN_NBUF + BUF_KLINV] = rows[s * GPRN_NBUF + BUF_X];
        }
        TRY(upload_table(c, c->tab_setup, rows));
        HIP_TRY(c, hipMemcpy(c->d_slotgp_setup, gps.data(), nb * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemsetAsync(c->d_info, 0, 3 * (size_t)c->nslot * sizeof(int), c->stream));
        c->d_ptrs = c->tab_setup;
        c->d_info_cur = c->d_info;
        TRY(factor_invert(c, nb, true));
        // log det K: non-owned helper entries are dropped below, before the all-reduce
        TRY(vec_logdet(c, BUF_B, c->d_slotgp_setup, nb, c->d_logdetK));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);      // (the verdicts are read through the null stream, which does not wait
        TRY(check_info(c, c->d_info, gps, &first_info));  // for the library's non-blocking streams by itself)
        // K_j^-1 = X^T X for the nodes that need it (one at a time: output goes to Kinv[j])
        for (int s = 0; s < nb; ++s) {
            const int g = gps[s];
            if (g >= c->q || !need_inv[g]) continue;
            if (!c->Kinv[g]) TRY(dev_alloc(c, &c->Kinv[g], nn));
            std::vector<double*> one((size_t)c->nslot * GPRN_NBUF, nullptr);
            one[BUF_B] = c->Kinv[g];
            one[BUF_X] = rows[s * GPRN_NBUF + BUF_X];
            HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
            TRY(upload_table(c, c->tab_setup, one));
            TRY(lauum_lower(c, 1));
            HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
            TRY(upload_table(c, c->tab_setup, rows));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    }
    // non-owned helper factorizations wrote logdetK[j] too: keep only owned entries, then share
    {
        std::vector<double> h(c->G);
        HIP_TRY(c, hipMemcpy(h.data(), c->d_logdetK, c->G * sizeof(double), hipMemcpyDeviceToHost));
        for (int g = 0; g < c->G; ++g) if (c->owner[g] != c->rank) h[g] = 0.0;
        HIP_TRY(c, hipMemcpy(c->d_logdetK, h.data(), c->G * sizeof(double), hipMemcpyHostToDevice));
        if (comm_active(c)) TRY(comm_allreduce(c, c->d_logdetK, c->G, false));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    }
    TRY(factor_check_waits(c));
    c->factored = true;
    return first_info;
}

// ------------------------------------------------------------------ sweep
// What of a phase's head and tail runs beside a factorisation (bits; option "overlap", default all):
//   1 B formed inside the first panel's update   2 row reductions over X panel by panel   4 node term beside the weight phase
//   8 log det B in k_finalize   16 the end of a sweep beside the next sweep's node phase
// (bit 32 of round 3 -- ... with the X^T X product of quirk Q1 too, the node phases alternating between two sets of
// workspaces -- measured 108.4 against 111.3 sweeps/s at config 3 and is gone: DESIGN.md 5d)
static int overlap_mask(const gprn_ctx* c) { return c->overlap_opt >= 0 ? c->overlap_opt : 31; }

static int mu_k_mu(gprn_ctx* c, bool weights, hipStream_t stream = nullptr, double* out = nullptr);

// One sweep of the small path (smalln.hip): node half-sweep, weight half-sweep, tail -- three launches, no host step
// between them.  (mu_in, var_in) is the state the sweep starts from, (mu_out, var_out) receives the new one.
static int small_sweep(gprn_ctx* c, const double* mu_in, const double* var_in, double* mu_out, double* var_out,
                       double* out4, double* scal, const SmallLoop* loop)
{
    c->d_scal = scal;
    c->d_logdetB = scal; c->d_trBinv = scal + c->G; c->d_muKmu = scal + 2 * (size_t)c->G; c->d_q1 = scal + 3 * (size_t)c->G;
    const int* done = loop ? loop->ctl : nullptr;
    TRY(ensure_small_sweep_tabs(c));
    c->d_ptrs = c->tab_node; c->slot0 = 0; c->d_info_cur = c->d_info + (size_t)c->nslot;
    TRY(small_phase(c, false, c->d_slotgp_node, (int)c->loc_nodes.size(), mu_in, var_in, mu_out, var_out, done));
    c->d_ptrs = c->tab_weight; c->slot0 = (int)c->loc_nodes.size(); c->d_info_cur = c->d_info + 2 * (size_t)c->nslot;
    TRY(small_phase(c, true, c->d_slotgp_weight, (int)c->loc_weights.size(), mu_in, var_in, mu_out, var_out, done));
    return small_tail(c, out4, scal, mu_out, var_out, loop);
}

// One half-sweep's factorisation with its head and tail, against c->d_ptrs / slot0 / d_info_cur (set by the caller): d, s,
// right-hand side -> B = I + D^1/2 K D^1/2 = L L^T, X = L^-1 -> u = X z, column sums over X -> the new rows of the state,
// tr B^-1, log det B.  `ns` slots whose latent GPs are d_slot_gp[slot] (and, for a batch of evaluations, whose evaluation
// is c->ev.slot_eval[slot]: midn.hip).
int phase_core(gprn_ctx* c, bool weights, const int* slotgp, int ns)
{
    const size_t o = (size_t)c->slot0 * c->ld;
    TRY(vec_prep(c, weights, slotgp, ns));
    // B = I + D^1/2 K D^1/2: built by factor_invert -- only the tiles its first outer panel's tile steps touch; the
    // others are formed from K inside that panel's K = 512 update (overlap bit 1).
    // The reductions over the rows of X = L^-1 (u = X z, column norms, X^T u: 8 N^2 bytes per matrix) run outer
    // panel by outer panel as the rows become final (rows_final, called by the schedule on the bulk stream: bit 2);
    // behind the factorisation only the last panel's rows, the reduction over the partial sums and the new state
    // are left.  Same kernels, same partial sums, same order of every addition: bit-identical results.
    const int overlap = overlap_mask(c);
    c->rows_done = 0;
    c->build_pending = ns;
    c->ft_s_phase = (overlap & 1) ? c->d_s + o : nullptr;
    if (overlap & 2) {
        c->rows_final = [c, o, slotgp, ns](int r0, int r1, hipStream_t st) -> int {
            TRY(vec_lower_matvec(c, BUF_X, c->d_z + o, c->ld, 0, slotgp, ns, c->d_u + o, st, r0 * GPRN_TILE,
                                 (r1 - r0) * GPRN_TILE));
            return vec_colops_partial(c, ns, st, r0, r1 - r0);
        };
    }
    const int rc_f = factor_invert(c, ns);
    c->ft_s_phase = nullptr; c->build_pending = 0;
    const int rd = c->rows_done;
    c->rows_final = nullptr; c->rows_done = 0;
    TRY(rc_f);
    TRY(vec_lower_matvec(c, BUF_X, c->d_z + o, c->ld, 0, slotgp, ns, c->d_u + o, nullptr, rd * GPRN_TILE, -1));
    TRY(vec_colops_partial(c, ns, nullptr, rd, -1));
    if (overlap & 8) TRY(vec_reduce_finalize(c, slotgp, ns, true));     // column sums, new state, tr B^-1, log det B
    else {
        TRY(vec_colops_reduce(c, ns));
        TRY(vec_logdet(c, BUF_B, slotgp, ns, c->d_logdetB));
        TRY(vec_finalize(c, slotgp, ns, false));
    }
    return GPRN_OK;
}

static int run_phase(gprn_ctx* c, bool weights)
{
    const std::vector<int>& gps = weights ? c->loc_weights : c->loc_nodes;
    const int ns = (int)gps.size();
    const int* slotgp = weights ? c->d_slotgp_weight : c->d_slotgp_node;
    c->d_ptrs = weights ? c->tab_weight : c->tab_node;
    c->slot0 = weights ? (int)c->loc_nodes.size() : 0;
    c->d_info_cur = c->d_info + (weights ? 2 : 1) * (size_t)c->nslot;
    const size_t o = (size_t)c->slot0 * c->ld;
    if (ns) {
        TRY(phase_core(c, weights, slotgp, ns));
        const int overlap = overlap_mask(c);
        if (c->keep_sigma) {
            const size_t nn = (size_t)c->ld * c->ld;
            TRY(lauum_lower(c, ns));
            for (int s = 0; s < ns; ++s) {
                if (!c->Sig[gps[s]]) {
                    TRY(dev_alloc(c, &c->Sig[gps[s]], nn));
                    HIP_TRY(c, hipMemsetAsync(c->Sig[gps[s]], 0, nn * sizeof(double), c->stream));   // padding stays zero
                }
                TRY(vec_sigma(c, c->wsB[c->slot0 + s], c->d_s + o + (size_t)s * c->ld, c->Sig[gps[s]]));
            }
        }
        if (!weights && c->q > 1) {
            // quirk Q1: <K_j^-1, Sigma_k> for k < j needs the explicit B_k^-1 = X^T X of every node
            // but the last.  Nothing in the weight phase reads it, so it runs beside that phase on
            // the second stream and is joined before the ELBO assembly.  It is handed to the weight
            // phase's factorisation, which enqueues it behind its first diagonal block (a launch of
            // 528 long-running workgroups just before would keep that block waiting for a free CU).
            const int n_inv = (gps.back() == c->q - 1) ? ns - 1 : ns;
            const std::vector<int> node_gps = gps;
            double** const node_tab = c->d_ptrs;
            const std::vector<double*> node_B(c->wsB.begin(), c->wsB.begin() + ns);
            double* const q1_out = c->d_q1;
            HIP_TRY(c, hipEventRecord(c->ev_nodes, c->stream));
            const bool early_term = (overlap & 4) && !c->loc_weights.empty();
            c->node_term_done = early_term;
            c->chain_started = [c, early_term, n_inv, ns, node_gps, node_tab, node_B, q1_out]() -> int {
                double** const cur = c->d_ptrs;
                const int cur_slot0 = c->slot0;
                HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_nodes, 0));
                // mu_f^T K_f^-1 mu_f needs the node phase's result only: HBM-bound work beside the MFMA-bound weight phase
                int rc = early_term ? mu_k_mu(c, false, c->stream2) : GPRN_OK;
                c->d_ptrs = node_tab;
                if (!rc && n_inv && !c->keep_sigma) rc = lauum_lower(c, n_inv, c->stream2);
                for (int s = 0; s < ns && !rc; ++s) {
                    const int k = node_gps[s];
                    for (int j = k + 1; j < c->q && !rc; ++j)
                        rc = vec_q1(c, c->Kinv[j], node_B[s], c->d_s + (size_t)s * c->ld, c->d_u,
                                    q1_out + (size_t)j * c->q + k, c->stream2);
                }
                c->d_ptrs = cur;
                c->slot0 = cur_slot0;
                if (rc) return rc;
                HIP_TRY(c, hipEventRecord(c->ev_q1, c->stream2));
                return GPRN_OK;
            };
            c->q1_pending = true;
        }
    }
    if (weights && c->chain_started) {
        // no factorisation took it along (no weight GP on this rank): now
        std::function<int()> f;
        f.swap(c->chain_started);
        TRY(f());
    }
    return exchange_rows(c, weights);
}

static int mu_k_mu(gprn_ctx* c, bool weights, hipStream_t stream, double* out)
{
    const std::vector<int>& gps = weights ? c->loc_weights : c->loc_nodes;
    const int ns = (int)gps.size();
    if (!ns) return GPRN_OK;
    const int* slotgp = weights ? c->d_slotgp_weight : c->d_slotgp_node;
    c->d_ptrs = weights ? c->tab_weight : c->tab_node;
    // a = L_K^-1 m_g with m_g = state row g (nodes: mu_f[g]; weights: the raw-reshape row, quirk Q2)
    c->slot0 = weights ? (int)c->loc_nodes.size() : 0;
    double* a = c->d_u + (size_t)c->slot0 * c->ld;
    TRY(vec_lower_matvec(c, BUF_KLINV, c->d_mu, c->N, 1, slotgp, ns, a, stream));
    return vec_dot_self(c, slotgp, ns, a, out ? out : c->d_muKmu, stream);
}

static int sweep_impl(gprn_ctx* c, int n_sweeps, int commit, double* elbo_out, double* parts_out, bool retry);

extern "C" int gprn_sweep(gprn_ctx* c, int n_sweeps, int commit, double* elbo_out, double* parts_out)
{
    DeviceLock lock_(c);
    WatchScope watch_(c, "gprn_sweep");
    if (!c) return GPRN_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int pre = GPRN_OK;
    if (n_sweeps <= 0 || !elbo_out) pre = bad(c, "sweep: bad argument");
    else if (!c->factored || !c->have_yres || !c->have_jit || !c->have_muvar)
        pre = bad(c, "sweep: needs factor_priors, set_y_resid, set_jitters and set_muvar first");
    if ((pre = agree_to_start(c, pre, "sweep"))) return pre;
    return with_event_fallback(c, "sweep", [&](bool retry) {
        return sweep_impl(c, n_sweeps, commit, elbo_out, parts_out, retry); }, true);
}

static int sweep_impl(gprn_ctx* c, int n_sweeps, int commit, double* elbo_out, double* parts_out, bool retry)
{
    if (n_sweeps > c->out_cap) {
        dev_free(c->d_out);
        TRY(dev_alloc(c, &c->d_out, 4 * (size_t)n_sweeps));
        c->out_cap = n_sweeps;
    }
    const size_t dn = (size_t)(c->p + 1) * c->q * c->N * sizeof(double);
    const bool small = small_applies(c);
    // the state the call started from: what commit = 0 returns to, and what a re-run starts over from (the small path
    // writes every new state into the OTHER copy: a committed call needs no snapshot, and it has nothing to re-run)
    if (small && commit) { /* nothing to keep */ }
    else if (retry) {
        HIP_TRY(c, hipMemcpyAsync(c->d_mu, c->d_mu_save, dn, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_var, c->d_var_save, dn, hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->d_mu_save, c->d_mu, dn, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_var_save, c->d_var, dn, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, 3 * (size_t)c->nslot * sizeof(int), c->stream));
    // The end of a sweep -- mu_w^T K_w^-1 mu_w (one pass over the six L_K^-1), the ELBO assembly and the wait for the
    // Q1 traces, some 150 us on the chain stream -- reads only what the sweep has left behind, and the next sweep's node
    // phase reads none of its results: inside a call of several sweeps it runs beside that phase, on the bulk stream,
    // handed to its factorisation like the X^T X product (chain_started: behind the first diagonal block, i.e. after
    // everything this sweep enqueued on the chain stream; the bulk stream is in order, so the Q1 traces are there too).
    // The per-GP scalars live in two copies for it.  Single-rank calls under the flag schedule only (no collective may
    // move; overlap bit 16); the last sweep of a call is assembled in line.
    const int overlap = overlap_mask(c);
    const size_t nscal = 3 * (size_t)c->G + (size_t)c->q * c->q;
    // (the node phase's factorisation must be one that joins the bulk stream at its end: an outer panel with a "rest")
    c->chain_started = nullptr;
    TRY(ensure_tasks(c));
    const int node_set = (int)c->loc_nodes.size() * c->T <= GPRN_LAT_MAX ? 1 : 0;
    const bool node_joins = !c->outers[node_set].empty() && c->outers[node_set][0].nrest > 0;
    const bool may_defer = (overlap & 16) && !comm_active(c) && factor_use_flags(c) == 1 &&
                           !c->loc_nodes.empty() && !c->loc_weights.empty() && !c->keep_sigma && node_joins;
    bool scal_cleared = small;                     // (the small path's kernels write every entry they read)
    if (!comm_active(c) && !small) {
        HIP_TRY(c, hipMemsetAsync(c->d_scal_base, 0, 2 * nscal * sizeof(double), c->stream));
        scal_cleared = true;
    }
    for (int it = 0; it < n_sweeps; ++it) {
        double* const scal = c->d_scal_base + (size_t)(it & 1) * nscal;
        double* const part = c->d_elbo_part + (size_t)(it & 1) * GPRN_ELBO_PART_DOUBLES;
        c->d_scal = scal;
        c->d_logdetB = scal; c->d_trBinv = scal + c->G; c->d_muKmu = scal + 2 * (size_t)c->G; c->d_q1 = scal + 3 * (size_t)c->G;
        // (every entry a sweep reads it has written itself, with '=': the two copies are cleared once per call, above; on a
        // sharded context the all-reduce leaves the other ranks' entries behind, so there it is cleared every sweep)
        if (comm_active(c) || !scal_cleared) HIP_TRY(c, hipMemsetAsync(scal, 0, nscal * sizeof(double), c->stream));
        c->node_term_done = false;
        if (small) {
            // three launches: the two half-sweeps read the state the sweep starts from and write the other copy
            TRY(small_sweep(c, c->d_mu, c->d_var, c->d_mu_alt, c->d_var_alt, c->d_out + 4 * (size_t)it, scal, nullptr));
            std::swap(c->d_mu, c->d_mu_alt);
            std::swap(c->d_var, c->d_var_alt);
            continue;
        }
        if (comm_active(c) && it > 0 && (it & 63) == 0) {
            // a long call on a sharded context: let the host see the device's progress now and then, so that the collective
            // watchdog's budget bounds a STALL (a rank that died) and not the legitimate length of the call
            HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
        }
        TRY(run_phase(c, false));
        TRY(run_phase(c, true));
        const bool defer = may_defer && it + 1 < n_sweeps;
        if (c->q1_pending && !defer) {          // the Q1 traces (and the node term) computed behind the weight phase
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_q1, 0));
        }
        c->q1_pending = false;
        const bool node_term = !c->node_term_done;
        double* const out4 = c->d_out + 4 * (size_t)it;
        if (defer) {
            c->chain_started = [c, scal, part, out4, node_term]() -> int {
                double** const cur = c->d_ptrs;
                const int cur_slot0 = c->slot0;
                int rc = node_term ? mu_k_mu(c, false, c->stream2, scal + 2 * (size_t)c->G) : GPRN_OK;
                if (!rc) rc = mu_k_mu(c, true, c->stream2, scal + 2 * (size_t)c->G);
                if (!rc) rc = vec_elbo(c, out4, scal, part, c->stream2);
                c->d_ptrs = cur; c->slot0 = cur_slot0;
                return rc;
            };
            continue;
        }
        if (node_term) TRY(mu_k_mu(c, false));
        TRY(mu_k_mu(c, true));
        TRY(reduce_scalars(c));
        TRY(vec_elbo(c, out4, scal, part));
    }
    if (!commit) {
        HIP_TRY(c, hipMemcpyAsync(c->d_mu, c->d_mu_save, dn, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_var, c->d_var_save, dn, hipMemcpyDeviceToDevice, c->stream));
    }
    std::vector<double> h(4 * (size_t)n_sweeps);
    std::vector<int> h_info;
    if (small) {                                    // (the pivot verdicts ride along: one synchronisation per call)
        h_info.resize(3 * (size_t)c->nslot);
        HIP_TRY(c, hipMemcpyAsync(h_info.data(), c->d_info, h_info.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    for (int it = 0; it < n_sweeps; ++it) {
        elbo_out[it] = h[4 * it];
        if (parts_out) for (int k = 0; k < 3; ++k) parts_out[3 * it + k] = h[4 * it + 1 + k];
    }
    int first = 0;
    c->info_gp = -1;
    if (small) {
        for (int ph = 1; ph <= 2; ++ph) {
            const std::vector<int>& gps = ph == 1 ? c->loc_nodes : c->loc_weights;
            for (size_t sl = 0; sl < gps.size(); ++sl) {
                const int v = h_info[(size_t)ph * c->nslot + sl];
                if (v > 0 && first == 0) { first = v; c->info_gp = gps[sl]; }
            }
        }
        return first;
    }
    TRY(factor_check_waits(c));
    TRY(check_info(c, c->d_info + (size_t)c->nslot, c->loc_nodes, &first));
    TRY(check_info(c, c->d_info + 2 * (size_t)c->nslot, c->loc_weights, &first));
    return first;
}

// ------------------------------------------------------------------ the ELBOcalc loop
extern "C" int gprn_factor_priors(gprn_ctx* c);
extern "C" int gprn_get_muvar(gprn_ctx* c, double* mu, double* var);
// meanfield.py:626-649 in one call: the first sweep's update is discarded and its ELBO kept as elboArray[0] (quirk Q7), then
// sweeps until `iterNumber > 3 and |std(last3) / mean(last3)| < 1e-3 and != 0` (np.std: population) or max_iter.
// the small path: ONE call, one synchronisation per batch of sweeps.  Inputs go through a pinned staging buffer and
// asynchronous copies, the set-up (fills + k_small_prior) is enqueued without waiting for its verdict, the loop runs on the
// device (k_small_tail applies the stop rule; sweeps enqueued ahead of the verdict become no-ops once it is in), and both
// copies of the state come back with the batch's read-back, so the final one is there whichever trip ended the loop.
struct ElboIo { int do_setup; const double *y_resid, *jitters, *mu, *var; double *mu_out, *var_out; };

static int elbocalc_small(gprn_ctx* c, const ElboIo& io, int max_iter, std::vector<double>& hist, int* iters, int* conv, int* info)
{
    const int K = 8;                                           // sweeps per batch
    static int stamps_env = -1;                                // GPRN_SMALL_STAMPS=1 (probes): where a half-sweep's time goes
    if (stamps_env < 0) { const char* e = getenv("GPRN_SMALL_STAMPS"); stamps_env = e ? atoi(e) : 0; }
    if (stamps_env && !c->d_small_stamps) {
        HIP_TRY(c, hipMalloc(&c->d_small_stamps, 8 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_small_stamps, 0, 8 * sizeof(unsigned long long)));
    }
    TRY(build_tables(c));
    const size_t pn = (size_t)c->p * c->N, d = (size_t)(c->p + 1) * c->q * c->N;
    const size_t n_info = 3 * (size_t)c->nslot;
    // pinned staging: in = y_resid | variance | mu | var;  out = A | Av | B | Bv | batch history | ctl (4 ints) | info
    const size_t in_doubles = 2 * pn + 2 * d, out_doubles = 4 * d + K + 4 + (n_info + 1) / 2 + 2;
    if (c->pin_in_cap < in_doubles) {
        if (c->h_pin_in) hipHostFree(c->h_pin_in);
        c->h_pin_in = nullptr; c->pin_in_cap = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_pin_in, in_doubles * sizeof(double), hipHostMallocDefault));
        c->pin_in_cap = in_doubles;
    }
    if (c->pin_out_cap < out_doubles) {
        if (c->h_pin_out) hipHostFree(c->h_pin_out);
        c->h_pin_out = nullptr; c->pin_out_cap = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_pin_out, out_doubles * sizeof(double), hipHostMallocDefault));
        c->pin_out_cap = out_doubles;
    }
    if (!c->d_loop_ctl) {
        HIP_TRY(c, hipMalloc(&c->d_loop_ctl, 4 * sizeof(int)));
        TRY(dev_alloc(c, &c->d_loop_hist, (size_t)K + 4));
    }
    // ---- inputs
    double* const pin = c->h_pin_in;
    if (io.y_resid) {
        memcpy(pin, io.y_resid, pn * sizeof(double));
        HIP_TRY(c, hipMemcpyAsync(c->d_yres, pin, pn * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_yres = true;
    }
    if (io.jitters) {
        double* v = pin + pn;
        for (int i = 0; i < c->p; ++i)
            for (int n = 0; n < c->N; ++n)
                v[(size_t)i * c->N + n] = io.jitters[i] * io.jitters[i] + c->h_yerr2[(size_t)i * c->N + n];
        HIP_TRY(c, hipMemcpyAsync(c->d_variance, v, pn * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_jit = true;
    }
    if (io.mu && io.var) {
        memcpy(pin + 2 * pn, io.mu, d * sizeof(double));
        memcpy(pin + 2 * pn + d, io.var, d * sizeof(double));
        HIP_TRY(c, hipMemcpyAsync(c->d_mu, pin + 2 * pn, d * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_var, pin + 2 * pn + d, d * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_muvar = true;
    }
    if (!c->have_yres || !c->have_jit || !c->have_muvar) return bad(c, "elbocalc: y_resid, jitters and the state must be given or set before");
    // (the pivot verdicts are raised by the kernels and cleared here, once per call: the set-up's row by the set-up)
    HIP_TRY(c, hipMemsetAsync(c->d_info + (io.do_setup ? (size_t)c->nslot : 0), 0, (io.do_setup ? 2 : 3) * (size_t)c->nslot * sizeof(int), c->stream));
    if (io.do_setup) TRY(factor_priors_small(c, false));
    else if (!c->factored) return bad(c, "elbocalc: no set-up yet (do_setup = 0)");
    HIP_TRY(c, hipMemsetAsync(c->d_loop_ctl, 0, 4 * sizeof(int), c->stream));
    double* const A = c->d_mu; double* const Av = c->d_var;
    double* const B = c->d_mu_alt; double* const Bv = c->d_var_alt;
    double* const scal = c->d_scal_base;
    if (c->out_cap < 1) { dev_free(c->d_out); TRY(dev_alloc(c, &c->d_out, 4)); c->out_cap = 1; }
    double* const po = c->h_pin_out;
    double* const hb = po + 4 * d;
    int* const ctl = reinterpret_cast<int*>(hb + K);
    int* const h_info = reinterpret_cast<int*>(hb + K + 2);
    // Quirk Q7: sweep 0 (the first ELBOaux call: update discarded, ELBO kept as elboArray[0], :627-628) and trip 1 are the
    // same computation on the same input -- it runs once, as trip 1, and its value is entered twice.  max_iter = 0 is the
    // one case that enqueues sweep 0.
    int s = max_iter >= 1 ? 1 : 0, iter = 0, done = 0;
    *conv = 0; *info = 0; c->info_gp = -1;
    hist.clear();
    while (!done && s <= max_iter) {
        const int s0 = s;
        int nb = 0;
        // (the stop rule cannot fire before trip 4, and a warm-started evaluation -- nELBO's case -- usually stops there: the
        // first batch ends at trip 4, so that no sweep is enqueued past the usual verdict; 4.7 us per no-op launch otherwise)
        const int nb_max = s0 <= 1 ? 4 : K;
        for (; nb < nb_max && s <= max_iter; ++nb, ++s) {
            // sweep 0 (discarded) and trip 1 both start from A; from then on the copies alternate
            const bool from_a = s <= 1 || (s & 1);
            SmallLoop loop{c->d_loop_ctl, c->d_loop_hist, c->d_loop_hist + K, s, nb, max_iter};
            TRY(small_sweep(c, from_a ? A : B, from_a ? Av : Bv, from_a ? B : A, from_a ? Bv : Av, c->d_out, scal, &loop));
        }
        HIP_TRY(c, hipMemcpyAsync(ctl, c->d_loop_ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(hb, c->d_loop_hist, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(h_info, c->d_info, n_info * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (io.mu_out && io.var_out) {
            HIP_TRY(c, hipMemcpyAsync(po, A, d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(po + d, Av, d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(po + 2 * d, B, d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(po + 3 * d, Bv, d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
        done = ctl[0];
        iter = ctl[1];
        *conv = ctl[2];
        const int ran = done ? std::min(nb, ctl[3] - s0 + 1) : nb;   // sweeps of the batch that were not no-ops
        if (s0 == 1 && ran > 0) hist.push_back(hb[0]);               // elboArray[0] == elboArray[1]
        for (int i = 0; i < ran; ++i) hist.push_back(hb[i]);
        for (int ph = 0; ph <= 2 && *info == 0; ++ph) {             // (row 0: the set-up's verdicts, slots = nodes then weights)
            if (ph == 0 && !io.do_setup) continue;
            for (int sl = 0; sl < c->nslot && *info == 0; ++sl) {
                const int v = h_info[(size_t)ph * c->nslot + sl];
                if (v <= 0) continue;
                const size_t nn_ = c->loc_nodes.size();
                int gp = -1;
                if (ph == 0) gp = (size_t)sl < nn_ ? c->loc_nodes[sl] : ((size_t)sl - nn_ < c->loc_weights.size() ? c->loc_weights[sl - nn_] : -1);
                else if (ph == 1) gp = (size_t)sl < nn_ ? c->loc_nodes[sl] : -1;
                else gp = (size_t)sl < c->loc_weights.size() ? c->loc_weights[sl] : -1;
                if (gp >= 0) { *info = v; c->info_gp = gp; }
            }
        }
    }
    if (c->d_small_stamps) {
        static int printed = 0;
        unsigned long long st[8];
        if (printed < 6 && hipMemcpy(st, c->d_small_stamps, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess && st[0]) {
            ++printed;
            fprintf(stderr, "[gprn] node half-sweep (small path), us: prep %.1f build %.1f factor %.1f publish %.1f matvec %.1f "
                            "colsums %.1f finalise %.1f | total %.1f\n", (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01,
                    (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01, (st[6] - st[5]) * 0.01,
                    (st[7] - st[6]) * 0.01, (st[7] - st[0]) * 0.01);
        }
    }
    *iters = iter;
    // the state the loop ended in: trip `iter` wrote it (A for even trips, B for odd ones; no trip: the state set by the caller)
    const bool in_b = iter >= 1 && (iter & 1);
    if (in_b) { c->d_mu = B; c->d_var = Bv; c->d_mu_alt = A; c->d_var_alt = Av; }
    if (io.mu_out && io.var_out) {
        memcpy(io.mu_out, po + (in_b ? 2 * d : 0), d * sizeof(double));
        memcpy(io.var_out, po + (in_b ? 3 * d : d), d * sizeof(double));
    }
    return GPRN_OK;
}

extern "C" int gprn_elbocalc(gprn_ctx* c, int do_setup, const double* y_resid, const double* jitters, const double* mu,
                             const double* var, int max_iter, double* history, int cap, int* n_history, int* iterations,
                             int* converged, double* mu_out, double* var_out)
{
    DeviceLock lock_(c);
    if (!c) return GPRN_E_ARG;
    WatchScope watch_(c, "gprn_elbocalc");
    // On a sharded context every LOCAL finding -- arguments, call order, a setter that fails -- goes into `pre`, and the
    // ranks agree on it before the first collective of the call (the set-up's own): a rank that returned here on its own
    // would leave the others in gprn_factor_priors' all-reduce.
    int pre = GPRN_OK;
    if (!c->N || max_iter < 0 || !history || cap < 1 || !n_history || !iterations || !converged || (!mu != !var) ||
        (!mu_out != !var_out))
        pre = bad(c, "elbocalc: bad argument");
    if (!pre && hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice"; pre = GPRN_E_HIP; }
    if (!pre && do_setup) {
        if (c->owner.empty()) pre = bad(c, "elbocalc: call set_owners first");
        for (int g = 0; g < c->G && !pre; ++g)
            if (!c->kspec[g].set) pre = bad(c, "elbocalc: a latent GP has no kernel");
    }
    std::vector<double> hist;
    int iter = 0, conv = 0, info = 0;
    if (!pre && small_applies(c)) {
        const ElboIo io{do_setup, y_resid, jitters, mu, var, mu_out, var_out};
        TRY(elbocalc_small(c, io, max_iter, hist, &iter, &conv, &info));
    } else {
        int rc;
        if (!pre && y_resid) pre = gprn_set_y_resid(c, y_resid);
        if (!pre && jitters) pre = gprn_set_jitters(c, jitters);
        if (!pre && mu) pre = gprn_set_muvar(c, mu, var);
        if (!pre && (!c->have_yres || !c->have_jit || !c->have_muvar || (!do_setup && !c->factored)))
            pre = bad(c, "elbocalc: needs the set-up, y_resid, jitters and the state (given or set before)");
        if ((pre = agree_to_start(c, pre, "elbocalc"))) return pre;
        if (do_setup) {
            rc = gprn_factor_priors(c);
            if (rc < 0) return rc;
            info = rc;
        }
        double e = 0.0;
        // Quirk Q7: the first ELBOaux call's update is discarded and its ELBO kept as elboArray[0] (:627-628); the loop's
        // first trip then repeats that very call (same state in, :636) -- elboArray[1] == elboArray[0] by construction.
        // The sweep is deterministic (no atomics in any reduction), so it runs ONCE, committed, and its ELBO is entered
        // twice; only max_iter = 0 needs the uncommitted form.
        // The stop rule cannot fire before trip 4 (:640), so trips 1 .. min(4, max_iter) go out as ONE call of sweep_impl: one
        // host synchronisation instead of four, and each sweep's ELBO assembly runs beside the next sweep's node phase
        // (overlap bit 16: same bits).  A warm-started evaluation -- nELBO's case -- usually stops right there.
        const int first_commit = max_iter >= 1 ? 1 : 0;
        const int nfirst = std::max(1, std::min(max_iter, 4));
        double efirst[4] = {0.0, 0.0, 0.0, 0.0};
        rc = with_event_fallback(c, "sweep", [&](bool retry) { return sweep_impl(c, nfirst, first_commit, efirst, nullptr, retry); }, true);
        if (rc < 0) return rc;
        if (!info) info = rc;
        e = efirst[0];
        hist.push_back(e);
        if (first_commit) {
            for (int k = 0; k < nfirst; ++k) hist.push_back(efirst[k]);
            iter = nfirst;
            const size_t n = hist.size();
            if (iter > 3 && elbo_stop_rule(hist[n - 3], hist[n - 2], hist[n - 1])) conv = 1;
        }
        while (!conv && iter < max_iter) {
            rc = with_event_fallback(c, "sweep", [&](bool retry) { return sweep_impl(c, 1, 1, &e, nullptr, retry); }, true);
            if (rc < 0) return rc;
            if (!info) info = rc;
            hist.push_back(e);
            iter += 1;
            const size_t n = hist.size();
            if (iter > 3 && elbo_stop_rule(hist[n - 3], hist[n - 2], hist[n - 1])) { conv = 1; break; }
        }
        if (mu_out && (rc = gprn_get_muvar(c, mu_out, var_out))) return rc;
    }
    *n_history = (int)hist.size();
    *iterations = iter;
    *converged = conv;
    // (a history longer than the caller's array keeps its LAST values: the first is elboArray[0] of a loop that ran to max_iter)
    const int n = (int)hist.size(), keep = std::min(n, cap);
    for (int i = 0; i < keep; ++i) history[i] = hist[(size_t)(n - keep) + i];
    return info;
}

// Device memory one chunk of side-by-side evaluations may take: option "batch_mem_mb", else half of what is free now, 48 GiB
// at most (the card holds 288: the rest stays with the caller's other contexts)
size_t batch_budget_bytes(gprn_ctx* c)
{
    if (c->batch_mem_mb > 0) return (size_t)c->batch_mem_mb << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    return std::min<size_t>(free_b / 2, (size_t)48 << 30);
}

// B independent evaluations of the loop above, side by side on the device: see include/gprn_hip.h.  One-tile problems run a
// half-sweep of ALL evaluations as one launch (smalln.hip); larger ones go through the launch schedule with
// batch = evaluations x latent GPs (midn.hip).  Either way a list longer than the memory budget holds runs chunk by chunk.
extern "C" int gprn_elbocalc_batch(gprn_ctx* c, int n_eval, const double* kernel_params, int n_kernel_params,
                                   const double* y_resid, const double* jitters, const double* mu, const double* var,
                                   int max_iter, double* elbo, int* iterations, int* converged, int* info,
                                   double* mu_out, double* var_out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || n_eval < 1 || !kernel_params || !y_resid || !jitters || !mu || !var || max_iter < 0 || !elbo ||
        !iterations || !converged || !info || (!mu_out != !var_out))
        return bad(c, "elbocalc_batch: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->owner.empty()) return bad(c, "elbocalc_batch: call set_owners first");
    if (comm_active(c) || c->world != 1) { c->err = "elbocalc_batch: one rank only (a pool of ranks splits the list itself)"; return GPRN_E_UNSUPPORTED; }
    TRY(build_tables(c));
    const bool small = c->T == 1 && small_applies(c);
    const size_t d = (size_t)(c->p + 1) * c->q * c->N, pn = (size_t)c->p * c->N;
    int chunk = small ? small_batch_chunk(c) : n_eval;             // (midn.hip sizes its own chunks: it knows what a matrix costs)
    if (small) c->last_batch_chunk = std::min(chunk, n_eval);
    for (int e0 = 0; e0 < n_eval;) {
        const int ne = std::min(chunk, n_eval - e0);
        const double* kp = kernel_params + (size_t)e0 * n_kernel_params;
        double* mo = mu_out ? mu_out + (size_t)e0 * d : nullptr;
        double* vo = var_out ? var_out + (size_t)e0 * d : nullptr;
        auto run = small ? small_batch_elbocalc : mid_batch_elbocalc;
        const int rc = run(c, ne, kp, n_kernel_params, y_resid + (size_t)e0 * pn, jitters + (size_t)e0 * c->p, mu + (size_t)e0 * d,
                           var + (size_t)e0 * d, max_iter, elbo + e0, iterations + e0, converged + e0, info + e0, mo, vo);
        if (rc == GPRN_E_NOMEM && small && ne > 1) {
            // the budget is an estimate: the device has less in one piece than it reports free -- the same chunk in halves
            // (nothing of it has run: the buffers are allocated before anything is enqueued)
            small_batch_free(c);
            chunk = std::max(1, ne / 2);
            c->last_batch_chunk = chunk;
            c->err.clear();
            continue;
        }
        if (rc) return rc;
        e0 += ne;
    }
    return GPRN_OK;
}

// ------------------------------------------------------------------ read-back
extern "C" int gprn_keep_sigma(gprn_ctx* c, int on)
{
    DeviceLock lock_(c);
    if (!c) return GPRN_E_ARG;
    c->keep_sigma = on != 0;
    return GPRN_OK;
}

extern "C" int gprn_get_matrix(gprn_ctx* c, int which, int gp, double* out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || gp < 0 || gp >= c->G || !out) return bad(c, "get_matrix: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const double* src = nullptr;
    if (which == GPRN_M_K) src = c->K[gp];
    else if (which == GPRN_M_KLINV) src = c->KLinv[gp];
    else if (which == GPRN_M_SIGMA) src = c->Sig[gp];
    else if (which == GPRN_M_BX || which == GPRN_M_BL) {
        // the sweep's own workspaces of this latent GP: every local GP has its (B, X) pair, nodes first (build_tables)
        const std::vector<int>& gps = gp < c->q ? c->loc_nodes : c->loc_weights;
        for (size_t sl = 0; sl < gps.size() && c->tables_ready; ++sl)
            if (gps[sl] == gp) {
                const size_t slot = (gp < c->q ? 0 : c->loc_nodes.size()) + sl;
                src = which == GPRN_M_BX ? c->wsX[slot] : c->wsB[slot];
            }
    }
    if (!src) return bad(c, "get_matrix: not available on this rank (or keep_sigma was off)");
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy2D(out, (size_t)c->N * sizeof(double), src, (size_t)c->ld * sizeof(double),
                           (size_t)c->N * sizeof(double), c->N, hipMemcpyDeviceToHost));
    if (which == GPRN_M_KLINV || which == GPRN_M_BX || which == GPRN_M_BL)      // strictly-upper tiles are scratch: report a clean lower factor
        for (int m = 0; m < c->N; ++m)
            for (int n = m + 1; n < c->N; ++n) out[(size_t)m * c->N + n] = 0.0;
    return GPRN_OK;
}

// per-GP scalars of the last sweep, as the ELBO assembly (k_elbo) read them: log det B [G], tr(B^-1) [G],
// m^T K^-1 m [G], the cumulative-trace terms < K_j^-1, Sigma_k > [q*q, entry j*q + k, k < j] -- for tests that
// recombine the entropy and the prior term on the host (tests/test_parity_gpu.py, config 5's shape)
extern "C" int gprn_get_scalars(gprn_ctx* c, double* out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !out) return bad(c, "get_scalars: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy(out, c->d_scal, (3 * (size_t)c->G + (size_t)c->q * c->q) * sizeof(double), hipMemcpyDeviceToHost));
    return GPRN_OK;
}

extern "C" int gprn_get_logdet_K(gprn_ctx* c, double* out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !out) return bad(c, "get_logdet_K: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy(out, c->d_logdetK, c->G * sizeof(double), hipMemcpyDeviceToHost));
    return GPRN_OK;
}

// ------------------------------------------------------------------ prediction
// Conditional mean / variance of every latent GP at new times, from the current variational
// state: replaces _gp.GP.prediction (_gp.py:107-138) as called by inference._Prediction
// (meanfield.py:1289-1381): cov = K + 1.25e-12 I + diag(var), sol = cov^-1 mu,
// mean* = K* sol, var*_i = k(t*_i,t*_i) + 1.25e-12 - |L^-1 K*_i|^2.  Here: fused fills,
// the blocked factor+inverse (X = L^-1), sol = X^T X mu, W^T = K* X^T by the tile kernel.
static int predict_impl(gprn_ctx* c, int ns, const double* tstar, double* mean_out, double* var_out);

// Host-evaluated matrices of latent GP `gp` for the next gprn_predict call with the same `ns`: what a user-defined
// covFunction subclass -- whose K reached the device through gprn_upload_K -- needs in place of the fused fills.
extern "C" int gprn_predict_upload(gprn_ctx* c, int gp, int ns, const double* K_tiny, const double* Kstar, const double* kss)
{
    DeviceLock lock_(c);
    if (!c || !c->N || gp < 0 || gp >= c->G || ns <= 0 || !K_tiny || !Kstar || !kss)
        return bad(c, "predict_upload: bad argument");
    if (c->owner.empty()) return bad(c, "predict_upload: call set_owners first");
    if (c->owner[gp] != c->rank) return GPRN_OK;                    // not needed on this rank
    gprn_ctx::PredStage& st = c->pred_stage[gp];
    st.ns = ns;
    st.K.assign(K_tiny, K_tiny + (size_t)c->N * c->N);
    st.Kstar.assign(Kstar, Kstar + (size_t)ns * c->N);
    st.kss.assign(kss, kss + ns);
    return GPRN_OK;
}

__global__ void k_add_to_diagonal(double* __restrict__ A, int ld, const double* __restrict__ v, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) A[(size_t)i * ld + i] += v[i];
}

extern "C" int gprn_predict(gprn_ctx* c, int ns, const double* tstar, double* mean_out, double* var_out)
{
    DeviceLock lock_(c);
    WatchScope watch_(c, "gprn_predict");
    if (!c || !c->N) return bad(c, "predict: bad argument");
    if (c->owner.empty()) return bad(c, "predict: call set_owners first");
    HIP_TRY(c, hipSetDevice(c->device));
    int pre = GPRN_OK;
    if (ns <= 0 || !tstar || !mean_out || !var_out) pre = bad(c, "predict: bad argument");
    else if (!c->have_muvar) pre = bad(c, "predict: set_muvar (or a sweep) first");
    else
        for (int g = 0; g < c->G && !pre; ++g) {
            if (c->owner[g] != c->rank) continue;
            if (!c->kspec[g].set) pre = bad(c, "predict: a latent GP has no kernel");
            else if (c->kspec[g].uploaded) {
                auto it = c->pred_stage.find(g);
                if (it == c->pred_stage.end() || it->second.ns != ns)
                    pre = bad(c, "predict: a host-evaluated kernel needs gprn_predict_upload (K, K*, k**) for this ns first");
            }
        }
    if ((pre = agree_to_start(c, pre, "predict"))) { c->pred_stage.clear(); return pre; }
    // (everything it factors is refilled from the kernel specs, the staged matrices and the variational state)
    const int rc = with_event_fallback(c, "predict", [&](bool) { return predict_impl(c, ns, tstar, mean_out, var_out); },
                                       true);
    c->pred_stage.clear();
    return rc;
}

static int predict_impl(gprn_ctx* c, int ns, const double* tstar, double* mean_out, double* var_out)
{
    TRY(build_tables(c));
    std::vector<int> gps = c->loc_nodes;
    gps.insert(gps.end(), c->loc_weights.begin(), c->loc_weights.end());
    const int nloc = (int)gps.size();
    for (int g : gps) {
        if (!c->kspec[g].set) return bad(c, "predict: a latent GP has no kernel");
        if (c->kspec[g].uploaded) {
            auto it = c->pred_stage.find(g);
            if (it == c->pred_stage.end() || it->second.ns != ns)
                return bad(c, "predict: a host-evaluated kernel needs gprn_predict_upload (K, K*, k**) for this ns first");
        }
    }
    const int ld = c->ld, N = c->N, T = c->T;
    const int ns_pad = ((ns + GPRN_TILE - 1) / GPRN_TILE) * GPRN_TILE;
    const size_t need = (size_t)ns_pad * ld;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    if (nloc && (c->predKs.size() != (size_t)c->nslot || c->pred_cap < need)) {
        for (auto& p : c->predKs) dev_free(p);
        for (auto& p : c->predWT) dev_free(p);
        c->predKs.assign(c->nslot, nullptr); c->predWT.assign(c->nslot, nullptr);
        for (int s = 0; s < c->nslot; ++s) {
            TRY(dev_alloc(c, &c->predKs[s], need));
            TRY(dev_alloc(c, &c->predWT[s], need));
        }
        c->pred_cap = need;
        if (c->tab_pred) tab_forget(c, c->tab_pred);       // (a null argument forgets EVERY table's host copy)
        dev_free(c->tab_pred); dev_free(c->d_slotgp_all);
        TRY(dev_alloc(c, &c->tab_pred, (size_t)c->nslot * GPRN_NBUF));
        TRY(dev_alloc(c, &c->d_slotgp_all, c->nslot));
    }
    double *d_ts = nullptr, *d_kss = nullptr, *d_mean = nullptr, *d_pvar = nullptr, *d_all = nullptr;
    TileTask* d_t = nullptr;
    int rc = GPRN_OK, first = 0;
    std::vector<double*> rows((size_t)c->nslot * GPRN_NBUF, nullptr);
    std::vector<int> staterow(nloc);
    std::vector<TileTask> tasks;
    std::vector<double> hm, hv, pad;
    const bool gather = comm_active(c);
    auto row_of = [&](int g) {
        if (g < c->q) return g;
        const int kk = g - c->q, j = kk / c->p, i = kk % c->p;
        return (1 + i) * c->q + j;
    };
#define PTRY(expr) do { rc = (expr); if (rc) goto done; } while (0)
#define PHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = GPRN_E_HIP; goto done; } } while (0)
    if (nloc) {
        PTRY(dev_alloc(c, &d_ts, ns));
        PTRY(dev_alloc(c, &d_kss, (size_t)nloc * ns_pad));
        PTRY(dev_alloc(c, &d_mean, (size_t)nloc * ns_pad));
        PTRY(dev_alloc(c, &d_pvar, (size_t)nloc * ns_pad));
        PHIP(hipMemcpy(d_ts, tstar, ns * sizeof(double), hipMemcpyHostToDevice));
        for (int s = 0; s < nloc; ++s) {
            rows[(size_t)s * GPRN_NBUF + BUF_B] = c->wsB[s];
            rows[(size_t)s * GPRN_NBUF + BUF_X] = c->wsX[s];
            rows[(size_t)s * GPRN_NBUF + BUF_K] = c->predKs[s];
            rows[(size_t)s * GPRN_NBUF + BUF_KLINV] = c->predWT[s];
            staterow[s] = row_of(gps[s]);
        }
        PTRY(upload_table(c, c->tab_pred, rows));
        PHIP(hipMemcpy(c->d_slotgp_all, staterow.data(), nloc * sizeof(int), hipMemcpyHostToDevice));
        for (int s = 0; s < nloc; ++s) {
            const KernelSpec& ks = c->kspec[gps[s]];
            if (!ks.uploaded) {
                PTRY(launch_fill(c, ks, c->wsB[s], 1.25e-12, c->d_var + (size_t)staterow[s] * N));
                PTRY(launch_fill_rect(c, ks, 1.25e-12, d_ts, ns, ns_pad, c->predKs[s], d_kss + (size_t)s * ns_pad));
                continue;
            }
            // the caller's matrices: K (identity padding) + diag(var), K* (zero padding), k**
            const gprn_ctx::PredStage& st = c->pred_stage[gps[s]];
            pad.assign((size_t)ld * ld, 0.0);
            for (int m = 0; m < ld; ++m) {
                if (m < N) memcpy(&pad[(size_t)m * ld], &st.K[(size_t)m * N], N * sizeof(double));
                else pad[(size_t)m * ld + m] = 1.0;
            }
            PHIP(hipMemcpy(c->wsB[s], pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_add_to_diagonal, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->wsB[s], ld,
                               c->d_var + (size_t)staterow[s] * N, N);
            PHIP(hipGetLastError());
            pad.assign(need, 0.0);
            for (int m = 0; m < ns; ++m) memcpy(&pad[(size_t)m * ld], &st.Kstar[(size_t)m * N], N * sizeof(double));
            PHIP(hipMemcpy(c->predKs[s], pad.data(), need * sizeof(double), hipMemcpyHostToDevice));
            pad.assign(ns_pad, 0.0);
            memcpy(pad.data(), st.kss.data(), ns * sizeof(double));
            PHIP(hipMemcpy(d_kss + (size_t)s * ns_pad, pad.data(), ns_pad * sizeof(double), hipMemcpyHostToDevice));
        }
        PHIP(hipMemsetAsync(c->d_info, 0, 3 * (size_t)c->nslot * sizeof(int), c->stream));
        c->d_ptrs = c->tab_pred;
        c->slot0 = 0;
        c->d_info_cur = c->d_info;
        PTRY(factor_invert(c, nloc, true));
        PTRY(vec_lower_matvec(c, BUF_X, c->d_mu, N, 1, c->d_slotgp_all, nloc, c->d_u));   // u = X mu
        PTRY(vec_colops(c, nloc));                                                          // ct = X^T u
        for (int bt = 0; bt < ns_pad / GPRN_TILE; ++bt)
            for (int at = 0; at < T; ++at)
                tasks.push_back(TileTask{(int64_t)bt * GPRN_TILE * ld + (int64_t)at * GPRN_TILE,
                                         (int64_t)bt * GPRN_TILE * ld, (int64_t)at * GPRN_TILE * ld,
                                         (at + 1) * GPRN_TILE, BUF_KLINV, BUF_K, BUF_X,
                                         tile_modes(CM_SET, 0, 0)});
        PTRY(dev_alloc(c, &d_t, tasks.size()));
        PHIP(hipMemcpyAsync(d_t, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice, c->stream));
        PTRY(launch_tiles(c, d_t, tasks.size(), c->d_ptrs, nloc, ld, GPRN_T_UPDATE));
        PTRY(vec_pred_rows(c, nloc, ns, ns_pad, c->d_ct, d_kss, d_mean, d_pvar));
    }
    if (gather) {
        // every rank ends up with every latent GP's rows: the owners' results travel as one grouped broadcast
        // (2 G messages of ns doubles); ranks that own nothing take part all the same
        PTRY(dev_alloc(c, &d_all, 2 * (size_t)c->G * ns));
        for (int s = 0; s < nloc; ++s) {
            PHIP(hipMemcpyAsync(d_all + (size_t)gps[s] * ns, d_mean + (size_t)s * ns_pad, ns * sizeof(double),
                                hipMemcpyDeviceToDevice, c->stream));
            PHIP(hipMemcpyAsync(d_all + ((size_t)c->G + gps[s]) * ns, d_pvar + (size_t)s * ns_pad, ns * sizeof(double),
                                hipMemcpyDeviceToDevice, c->stream));
        }
        if (c->comm) { if (g_rccl.GroupStart() != ncclSuccess) { c->err = "ncclGroupStart"; rc = GPRN_E_COMM; goto done; } }
        for (int g = 0; g < c->G && !rc; ++g) {
            rc = comm_broadcast(c, d_all + (size_t)g * ns, ns, c->owner[g]);
            if (!rc) rc = comm_broadcast(c, d_all + ((size_t)c->G + g) * ns, ns, c->owner[g]);
        }
        if (c->comm) { if (g_rccl.GroupEnd() != ncclSuccess && !rc) { c->err = "ncclGroupEnd"; rc = GPRN_E_COMM; } }
        if (rc) goto done;
        PHIP(hipMemcpyAsync(mean_out, d_all, (size_t)c->G * ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PHIP(hipMemcpyAsync(var_out, d_all + (size_t)c->G * ns, (size_t)c->G * ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PHIP(hipStreamSynchronize(c->stream)); watch_progress(c);
    } else if (nloc) {
        hm.resize((size_t)nloc * ns_pad); hv.resize((size_t)nloc * ns_pad);
        PHIP(hipMemcpyAsync(hm.data(), d_mean, hm.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PHIP(hipMemcpyAsync(hv.data(), d_pvar, hv.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PHIP(hipStreamSynchronize(c->stream)); watch_progress(c);
        for (int s = 0; s < nloc; ++s) {
            memcpy(mean_out + (size_t)gps[s] * ns, &hm[(size_t)s * ns_pad], ns * sizeof(double));
            memcpy(var_out + (size_t)gps[s] * ns, &hv[(size_t)s * ns_pad], ns * sizeof(double));
        }
    }
    c->info_gp = -1;
    rc = factor_check_waits(c);
    if (!rc && nloc) rc = check_info(c, c->d_info, gps, &first);
    if (!rc) rc = first;
done:
#undef PTRY
#undef PHIP
    hipStreamSynchronize(c->stream);
    dev_free(d_ts); dev_free(d_kss); dev_free(d_mean); dev_free(d_pvar); dev_free(d_t); dev_free(d_all);
    return rc;
}

// ------------------------------------------------------------------ kernel matrices, prior samples
static int spec_from_args(gprn_ctx* c, KernelSpec& ks, const int32_t* ops, int n_ops, const double* params,
                          int n_params, int add_nugget)
{
    if (!ops || n_ops <= 0 || n_ops > GPRN_MAX_OPS || n_params < 0 || n_params > GPRN_MAX_KPARAMS || (n_params && !params))
        return bad(c, "kernel expression: bad argument");
    int depth = 0;
    for (int o = 0; o < n_ops; ++o) {
        const int op = ops[3 * o], kid = ops[3 * o + 1], off = ops[3 * o + 2];
        if (op == GPRN_OP_PUSH) {
            if (kid < 0 || kid >= GPRN_K_COUNT || off < 0 || off > n_params || ++depth > 8) return bad(c, "kernel expression: bad push");
        } else if (op == GPRN_OP_ADD || op == GPRN_OP_MUL) {
            if (--depth < 1) return bad(c, "kernel expression: malformed");
        } else return bad(c, "kernel expression: unknown opcode");
    }
    if (depth != 1) return bad(c, "kernel expression: malformed");
    ks.set = true; ks.uploaded = false;
    ks.n_ops = n_ops; ks.n_params = n_params; ks.nugget = add_nugget ? 1 : 0;
    memcpy(ks.ops, ops, 3 * n_ops * sizeof(int32_t));
    if (n_params) memcpy(ks.params, params, n_params * sizeof(double));
    return GPRN_OK;
}

static int test_setup(gprn_ctx* c, int ld, int nbuf_needed, int batch);

// K = expr(t_i, t_j) + nugget I at the data times, evaluated by the fused fill kernel: inference._KMatrix
// (meanfield.py:413-434, nugget 1e-6) and _tinyNuggetKMatrix (:436-452, 1.25e-12); nugget = 0 for the
// two-argument kernels.  K_out: (N, N) host.
extern "C" int gprn_eval_kernel(gprn_ctx* c, const int32_t* ops, int n_ops, const double* params, int n_params,
                                double nugget, double* K_out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !K_out) return bad(c, "eval_kernel: call set_data first");
    HIP_TRY(c, hipSetDevice(c->device));
    KernelSpec ks;
    TRY(spec_from_args(c, ks, ops, n_ops, params, n_params, nugget != 0.0));
    TRY(test_setup(c, c->ld, 1, 1));
    TRY(launch_fill(c, ks, c->d_test[0], nugget));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipMemcpy2D(K_out, (size_t)c->N * sizeof(double), c->d_test[0], (size_t)c->ld * sizeof(double),
                           (size_t)c->N * sizeof(double), c->N, hipMemcpyDeviceToHost));
    return GPRN_OK;
}

// Draws from the GP prior of a kernel at the data times: out[s] = L z[s] with K + nugget I = L L^T from the
// blocked factorisation (inference._sample_from_gp, meanfield.py:517-531, which hands K to
// scipy.stats.multivariate_normal).  z: (n_samples, N) standard normals from the caller's generator; a
// positive return is the LAPACK-style info of a K that is not positive definite at this nugget.
static int sample_prior_impl(gprn_ctx* c, const KernelSpec& ks, double nugget, int n_samples, const double* z,
                             double* out)
{
    const int ld = c->ld, N = c->N;
    TRY(test_setup(c, ld, 2, 1));
    double **d_p = nullptr, *d_z = nullptr, *d_o = nullptr;
    int* d_i = nullptr;
    int rc = dev_alloc(c, &d_p, GPRN_NBUF);
    if (!rc) rc = dev_alloc(c, &d_i, 1);
    if (!rc) rc = dev_alloc(c, &d_z, (size_t)n_samples * ld);
    if (!rc) rc = dev_alloc(c, &d_o, (size_t)n_samples * ld);
    double** const sptrs = c->d_ptrs;
    int* const sinfo = c->d_info_cur;
    int info0 = 0;
    hipError_t e = hipSuccess;
    if (!rc) {
        double* hp[GPRN_NBUF] = {c->d_test[0], c->d_test[1], nullptr, nullptr};
        e = hipMemcpy(d_p, hp, sizeof(hp), hipMemcpyHostToDevice);
        if (e == hipSuccess) tab_note(c, d_p, hp, GPRN_NBUF);
        if (e == hipSuccess) e = hipMemset(d_i, 0, sizeof(int));
        if (e == hipSuccess) e = hipMemset(d_z, 0, (size_t)n_samples * ld * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy2D(d_z, (size_t)ld * sizeof(double), z, (size_t)N * sizeof(double),
                                             (size_t)N * sizeof(double), n_samples, hipMemcpyHostToDevice);
        if (e == hipSuccess) rc = launch_fill(c, ks, c->d_test[0], nugget);
        c->d_ptrs = d_p; c->d_info_cur = d_i;
        if (e == hipSuccess && !rc) rc = factor_invert(c, 1, true);
        for (int s = 0; s < n_samples && e == hipSuccess && !rc; ++s)     // L z: row i of lower(B) . z
            rc = vec_lower_matvec(c, BUF_B, d_z + (size_t)s * ld, 0, 0, nullptr, 1, d_o + (size_t)s * ld);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && !rc) rc = factor_check_waits(c);
        if (e == hipSuccess && !rc) e = hipMemcpy(&info0, d_i, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess && !rc)
            e = hipMemcpy2D(out, (size_t)N * sizeof(double), d_o, (size_t)ld * sizeof(double),
                            (size_t)N * sizeof(double), n_samples, hipMemcpyDeviceToHost);
    }
    c->d_ptrs = sptrs; c->d_info_cur = sinfo;
    if (d_p) { tab_forget(c, d_p); hipFree(d_p); }
    if (d_i) hipFree(d_i);
    if (d_z) hipFree(d_z);
    if (d_o) hipFree(d_o);
    if (rc) return rc;
    HIP_TRY(c, e);
    return info0;
}

extern "C" int gprn_sample_prior(gprn_ctx* c, const int32_t* ops, int n_ops, const double* params, int n_params,
                                 double nugget, int n_samples, const double* z, double* out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || n_samples <= 0 || !z || !out) return bad(c, "sample_prior: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    KernelSpec ks;
    TRY(spec_from_args(c, ks, ops, n_ops, params, n_params, nugget != 0.0));
    TRY(ensure_tasks(c));
    return with_event_fallback(c, "sample_prior", [&](bool) { return sample_prior_impl(c, ks, nugget, n_samples, z, out); });
}

// ------------------------------------------------------------------ gradient pieces
// At fixed variational state only the expected log prior depends on the hyper-parameters of latent GP g's
// kernel (meanfield.py:992-1067):  -1/2 log det K - 1/2 (m^T K^-1 m + tr(K^-1 S)),  S = the covariance the
// reference pairs with K_g (node j: Sigma_f0 + ... + Sigma_fj, quirk Q1; weight: its own Sigma_w), so
//     d/dtheta = 1/2 < K^-1 S K^-1 + a a^T - K^-1 , dK/dtheta >,   a = K^-1 m.
// The N^3 part is done here, on the tile kernel: K^-1 = L_K^-T L_K^-1 and P = K^-1 S K^-1 for one latent GP,
// from the factors of gprn_factor_priors and the explicit Sigma of the last sweep (gprn_keep_sigma).  The
// O(N^2) contraction with dK/dtheta stays with the caller, who owns the kernel classes.
// Kinv_out, P_out: (N, N), both symmetric (full).  One rank only (the node sum needs every node's Sigma).
// kernel_grad != NULL: contract on the device instead of copying the matrices out -- needs a single SE / Periodic
// / QuasiPeriodic kernel on latent GP `gp` and its mean vector m (N); kernel_grad[l], l < n_params.
static int grad_impl(gprn_ctx* c, int gp, double* Kinv_out, double* P_out, const double* m, double* kernel_grad,
                     bool closed_form = false)
{
    if (c->world != 1) return bad(c, "grad_matrices: not available on a sharded context");
    if (!c->factored || !c->keep_sigma) return bad(c, "grad_matrices: needs factor_priors and a sweep with keep_sigma");
    const int nsum = gp < c->q ? gp + 1 : 1;
    for (int k = 0; k < nsum; ++k)
        if (!c->Sig[gp < c->q ? k : gp]) return bad(c, "grad_matrices: no Sigma yet (run a sweep with keep_sigma on)");
    if (c->nslot < 2) return bad(c, "grad_matrices: needs two workspace slots");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream2));
    const int ld = c->ld, N = c->N, T = c->T;
    const size_t nn = (size_t)ld * ld;
    // workspaces of the sweep are free between calls: slot 0's B holds K^-1, its X the sum S, slot 1's B the
    // product -K^-1 S, and P lands in slot 0's X once S has been read
    double* const dKinv = c->wsB[0];
    double* const dS = c->wsX[0];
    double* const dC1 = c->wsB[1];
    // S (full, ld x ld, padding zero)
    HIP_TRY(c, hipMemsetAsync(dS, 0, nn * sizeof(double), c->stream));
    for (int k = 0; k < nsum; ++k)
        TRY(vec_axpy_matrix(c, c->Sig[gp < c->q ? k : gp], dS, N));
    TileTask* d_t = nullptr;
    double** d_p = nullptr;
    std::vector<TileTask> tasks;
    auto toff = [&](int ti, int tj) { return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE; };
    // buffer slots of these launches: 0 = K^-1 (BUF_B), 1 = L_K^-1 (BUF_X, for the X^T X list), 2 = S then P, 3 = C1
    double* hp[GPRN_NBUF] = {dKinv, c->KLinv[gp], dS, dC1};
    int rc = dev_alloc(c, &d_p, GPRN_NBUF);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpy(d_p, hp, sizeof(hp), hipMemcpyHostToDevice);
    // (1) K^-1 = lower(X^T X), X = L_K^-1: the X^T X task list (BUF_X -> BUF_B); then mirror it to the upper
    // triangle so that the two products below read plain full tiles
    if (!rc && e == hipSuccess) rc = ensure_tasks(c);
    double** const sptrs = c->d_ptrs;
    c->d_ptrs = d_p;
    if (!rc && e == hipSuccess) rc = lauum_lower(c, 1);
    if (!rc && e == hipSuccess) rc = vec_symmetrize(c, dKinv);
    // (2) C1 = -K^-1 S, all T x T tiles, K = ld
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < T; ++j)
            tasks.push_back(TileTask{toff(i, j), toff(i, 0), toff(0, j), ld, 3, 0, 2, tile_modes(CM_SETNEG, 0, 1)});
    const size_t n1 = tasks.size();
    // (3) P = -C1 K^-1 = K^-1 S K^-1, into slot 2 (S is dead by then)
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < T; ++j)
            tasks.push_back(TileTask{toff(i, j), toff(i, 0), toff(0, j), ld, 2, 3, 0, tile_modes(CM_SETNEG, 0, 1)});
    if (!rc && e == hipSuccess) rc = dev_alloc(c, &d_t, tasks.size());
    if (!rc && e == hipSuccess)
        e = hipMemcpyAsync(d_t, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice, c->stream);
    if (!rc && e == hipSuccess) rc = launch_tiles(c, d_t, n1, d_p, 1, ld, GPRN_T_UPDATE);
    if (!rc && e == hipSuccess) rc = launch_tiles(c, d_t + n1, tasks.size() - n1, d_p, 1, ld, GPRN_T_UPDATE);
    c->d_ptrs = sptrs;
    if (kernel_grad) {
        // slot 1's X workspace is free: [0, ld) the mean vector, [ld, 2 ld) a = K^-1 m, then the per-row partial sums
        const KernelSpec& ks = c->kspec[gp];
        double* const w = c->wsX[1];
        double gh[GPRN_MAX_KPARAMS] = {0};
        const int np_out = closed_form ? 4 : ks.n_params;
        if (!rc && e == hipSuccess) e = hipMemcpyAsync(w, m, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (!rc && e == hipSuccess) {
            if (closed_form)
                rc = vec_grad_contract(c, ks.ops[1], ks.params, dKinv, dS, w, w + ld, w + 2 * (size_t)ld, w + 6 * (size_t)ld);
            else {
                rc = vec_symv(c, dKinv, w, w + ld);
                if (!rc) rc = launch_grad_fd(c, ks, dKinv, dS, w + ld, w + 2 * (size_t)ld, w + 6 * (size_t)ld);
            }
        }
        if (!rc && e == hipSuccess)
            e = hipMemcpyAsync(gh, w + 6 * (size_t)ld, (size_t)np_out * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        for (int l = 0; l < ks.n_params && l < np_out; ++l) kernel_grad[l] = gh[l];
    } else {
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (!rc && e == hipSuccess)
            e = hipMemcpy2D(Kinv_out, (size_t)N * sizeof(double), dKinv, (size_t)ld * sizeof(double),
                            (size_t)N * sizeof(double), N, hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess)
            e = hipMemcpy2D(P_out, (size_t)N * sizeof(double), dS, (size_t)ld * sizeof(double),
                            (size_t)N * sizeof(double), N, hipMemcpyDeviceToHost);
    }
    if (d_t) hipFree(d_t);
    if (d_p) { tab_forget(c, d_p); hipFree(d_p); }
    if (rc) return rc;
    HIP_TRY(c, e);
    return GPRN_OK;
}

extern "C" int gprn_grad_matrices(gprn_ctx* c, int gp, double* Kinv_out, double* P_out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || gp < 0 || gp >= c->G || !Kinv_out || !P_out) return bad(c, "grad_matrices: bad argument");
    return grad_impl(c, gp, Kinv_out, P_out, nullptr, nullptr);
}

// The whole kernel-parameter gradient of latent GP `gp` on the device: < 1/2 (K^-1 S K^-1 + a a^T - K^-1), dK/dtheta_l >,
// a = K^-1 m -- closed-form dK/dtheta for a single SquaredExponential, Periodic or QuasiPeriodic (csrc/vecops.hip), the
// central difference of the kernel program itself for every other built-in and Sum / Multiplication tree
// (csrc/fill.hip, launch_grad_fd); GPRN_E_ARG for a latent GP whose K was uploaded (user kernels: the caller then
// contracts gprn_grad_matrices' output itself).  m: the mean the reference pairs with that kernel (N); grad_out:
// n_params values (NOT yet divided by q).
extern "C" int gprn_grad_kernel(gprn_ctx* c, int gp, const double* m, double* grad_out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || gp < 0 || gp >= c->G || !m || !grad_out) return bad(c, "grad_kernel: bad argument");
    const KernelSpec& ks = c->kspec[gp];
    if (!ks.set || ks.uploaded || ks.n_ops < 1) {
        c->err = "grad_kernel: the kernel of this latent GP has no device program (uploaded matrix)";
        return GPRN_E_UNSUPPORTED;
    }
    const int kid = (ks.n_ops == 1 && ks.ops[0] == GPRN_OP_PUSH && ks.ops[2] == 0) ? ks.ops[1] : -1;
    const bool closed = kid == GPRN_K_SE || kid == GPRN_K_PERIODIC || kid == GPRN_K_QP;
    if (c->ld < 8 + GPRN_MAX_KPARAMS / 8) return bad(c, "grad_kernel: problem too small");
    return grad_impl(c, gp, nullptr, nullptr, m, grad_out, closed);
}

// ------------------------------------------------------------------ the ELBO's terms on their own
// inference._expectedLogLike (meanfield.py:895-990) of the state last set (gprn_set_muvar: the variances ARE the diagonals of
// Sigma_f / Sigma_w that the reference extracts, :688-697, 956-987) under the jitters last set: the same kernel the sweep's
// ELBO assembly uses (k_loglike_partial), its 32 partial sums added in k_elbo_final's order.
extern "C" int gprn_expected_loglike(gprn_ctx* c, double* logl_out)
{
    DeviceLock lock_(c);
    if (!c || !c->N || !logl_out) return bad(c, "expected_loglike: bad argument");
    if (!c->have_jit || !c->have_muvar) return bad(c, "expected_loglike: set_jitters and set_muvar first");
    HIP_TRY(c, hipSetDevice(c->device));
    // (scal is not read by the launch we keep: a throw-away ELBO assembly over whatever the scalars hold)
    double* part = c->d_elbo_part;
    if (c->out_cap < 1) { dev_free(c->d_out); TRY(dev_alloc(c, &c->d_out, 4)); c->out_cap = 1; }
    TRY(vec_elbo(c, c->d_out, c->d_scal_base, part));
    double h[GPRN_ELBO_PART_DOUBLES];
    HIP_TRY(c, hipMemcpyAsync(h, part, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    for (int b = 0; b < GPRN_ELBO_PART_DOUBLES / 3; ++b) { t1 += h[3 * b]; t2 += h[3 * b + 1]; t3 += h[3 * b + 2]; }
    *logl_out = -0.5 * t1 - 0.5 * t2 - 0.5 * t3;
    return GPRN_OK;
}

// out[i] = sum_{n <= i} A[i][n] W[i][n] over the lower triangle of two ld-pitched matrices (one wave per row)
__global__ __launch_bounds__(256)
void k_rowdot_lower(const double* __restrict__ A, const double* __restrict__ W, int N, int ld, double* __restrict__ out)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    double acc = 0.0;
    for (int n = lane; n <= i; n += 64) acc += A[(size_t)i * ld + n] * W[(size_t)i * ld + n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) out[i] = acc;
}

// What inference._expectedLogPrior (meanfield.py:992-1067) needs of latent GP `gp` for a covariance S and a mean m that the
// CALLER supplies (the reference pairs node j with the cumulative Sigma_f0 + ... + Sigma_fj and weight (j, i) with the
// raw-reshape row of mu_w: quirks Q1, Q2 -- the caller's business), from the factor of K_gp that gprn_factor_priors left on
// the device:  out[0] = log det K = 2 sum log diag chol(K) (:1029, 1062),  out[1] = m^T K^-1 m = |L^-1 m|^2 (:1032, 1050),
// out[2] = tr(K^-1 S) = < L^-1, L^-1 S > (:1041, 1051; the reference: cho_solve of the N x N matrix, 2 N^3 -- here one
// triangular product on the tile kernel, N^3).  S: (N, N), m: (N).  Unsharded contexts.
extern "C" int gprn_prior_terms(gprn_ctx* c, int gp, const double* S, const double* m, double* out3)
{
    DeviceLock lock_(c);
    if (!c || !c->N || gp < 0 || gp >= c->G || !S || !m || !out3) return bad(c, "prior_terms: bad argument");
    if (c->world != 1) return bad(c, "prior_terms: not available on a sharded context");
    if (!c->factored) return bad(c, "prior_terms: needs factor_priors first");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(build_tables(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream2));
    const int ld = c->ld, N = c->N, T = c->T;
    // the latent GP's own row of the phase tables: BUF_B <- S (zero padding), BUF_X <- W = L_K^-1 S, BUF_KLINV = L_K^-1
    const bool node = gp < c->q;
    const std::vector<int>& gps = node ? c->loc_nodes : c->loc_weights;
    int slot = -1;
    for (size_t sl = 0; sl < gps.size(); ++sl) if (gps[sl] == gp) slot = (int)sl;
    if (slot < 0) return bad(c, "prior_terms: latent GP not held here");
    double** const tab = (node ? c->tab_node : c->tab_weight) + (size_t)slot * GPRN_NBUF;
    const size_t ws = (node ? 0 : c->loc_nodes.size()) + (size_t)slot;
    double* const dS = c->wsB[ws];
    double* const dW = c->wsX[ws];
    HIP_TRY(c, hipMemsetAsync(dS, 0, (size_t)ld * ld * sizeof(double), c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(dS, (size_t)ld * sizeof(double), S, (size_t)N * sizeof(double), (size_t)N * sizeof(double), N,
                                hipMemcpyHostToDevice, c->stream));
    std::vector<TileTask> tasks;
    auto toff = [&](int ti, int tj) { return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE; };
    for (int ti = 0; ti < T; ++ti)                     // W(ti, tj) = sum_{k <= ti} L^-1(ti, k) S(k, tj): the factor is lower triangular
        for (int tj = 0; tj < T; ++tj)
            tasks.push_back(TileTask{toff(ti, tj), toff(ti, 0), toff(0, tj), (ti + 1) * GPRN_TILE, BUF_X, BUF_KLINV, BUF_B,
                                     tile_modes(CM_SET, 0, 1)});
    TileTask* d_t = nullptr;
    double* d_m = nullptr;
    int rc = dev_alloc(c, &d_t, tasks.size());
    if (!rc) rc = dev_alloc(c, &d_m, 3 * (size_t)ld + 4);
    hipError_t e = hipSuccess;
    double** const sptrs = c->d_ptrs;
    const int sslot0 = c->slot0;
    const EvalMap sev = c->ev;
    if (!rc) e = hipMemcpyAsync(d_t, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice, c->stream);
    if (!rc && e == hipSuccess) rc = launch_tiles(c, d_t, tasks.size(), tab, 1, ld, GPRN_T_UPDATE);
    if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(k_rowdot_lower, dim3((N + 3) / 4), dim3(256), 0, c->stream, (const double*)c->KLinv[gp], (const double*)dW,
                           N, ld, d_m + ld);
        e = hipGetLastError();
    }
    // tr(K^-1 S): the rows' sums in a fixed order; m^T K^-1 m: a = L^-1 m (one wave per row), then a . a
    double h[3] = {0.0, 0.0, 0.0};
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(d_m, m, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (!rc && e == hipSuccess) {
        static const int zero = 0;
        int* d_zero = nullptr;
        rc = dev_alloc(c, &d_zero, 1);
        if (!rc) e = hipMemcpyAsync(d_zero, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream);
        c->d_ptrs = tab; c->slot0 = 0; c->ev = EvalMap{nullptr, 0, 0, 0, 0};
        if (!rc && e == hipSuccess) rc = vec_lower_matvec(c, BUF_KLINV, d_m, 0, 0, d_zero, 1, d_m + 2 * (size_t)ld);
        // (one slot, "latent GP 0": the scalar lands at d_m[3 ld])
        if (!rc && e == hipSuccess) rc = vec_dot_self(c, d_zero, 1, d_m + 2 * (size_t)ld, d_m + 3 * (size_t)ld);
        if (!rc && e == hipSuccess) {
            std::vector<double> rows(N);
            e = hipMemcpyAsync(rows.data(), d_m + ld, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&h[1], d_m + 3 * (size_t)ld, sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&h[0], c->d_logdetK + gp, sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            for (int i = 0; i < N; ++i) h[2] += rows[i];
        }
        c->d_ptrs = sptrs; c->slot0 = sslot0; c->ev = sev;
        if (d_zero) hipFree(d_zero);
    }
    if (d_t) hipFree(d_t);
    if (d_m) hipFree(d_m);
    if (rc) return rc;
    if (e != hipSuccess) { c->err = std::string("prior_terms: ") + hipGetErrorString(e); return GPRN_E_HIP; }
    out3[0] = h[0]; out3[1] = h[1]; out3[2] = h[2];
    return GPRN_OK;
}

// ------------------------------------------------------------------ diagnostics
static int test_setup(gprn_ctx* c, int ld, int nbuf_needed, int batch)
{
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    const size_t nn = (size_t)ld * ld * batch;
    for (int b = 0; b < 3; ++b) {
        if (b < nbuf_needed && c->test_cap[b] < nn) {
            dev_free(c->d_test[b]);
            TRY(dev_alloc(c, &c->d_test[b], nn));
            c->test_cap[b] = nn;
        }
    }
    return GPRN_OK;
}

extern "C" int gprn_test_gemm(gprn_ctx* c, int M, int N, int K, int a_mode, int b_mode, int c_mode,
                              const double* A, const double* B, double* C)
{
    DeviceLock lock_(c);
    if (!c || M <= 0 || N <= 0 || K <= 0 || M % GPRN_TILE || N % GPRN_TILE || K % GPRN_KC || !A || !B || !C)
        return bad(c, "test_gemm: bad argument");
    const int ld = std::max(std::max(M, N), K);
    TRY(test_setup(c, ld, 3, 1));
    // place the operands in ld x ld row-major buffers exactly as the task modes address them
    const size_t nn = (size_t)ld * ld;
    std::vector<double> ha(nn, 0.0), hb(nn, 0.0), hc(nn, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const double v = A[(size_t)m * K + k];
            if (a_mode == 0) ha[(size_t)m * ld + k] = v; else ha[(size_t)k * ld + m] = v;
        }
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) {
            const double v = B[(size_t)k * N + n];
            if (b_mode == 0) hb[(size_t)n * ld + k] = v; else hb[(size_t)k * ld + n] = v;
        }
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) hc[(size_t)m * ld + n] = C[(size_t)m * N + n];
    HIP_TRY(c, hipMemcpy(c->d_test[0], ha.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_test[1], hb.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_test[2], hc.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    std::vector<TileTask> tasks;
    for (int ti = 0; ti < M / GPRN_TILE; ++ti)
        for (int tj = 0; tj < N / GPRN_TILE; ++tj) {
            TileTask t;
            t.c_off = (int64_t)ti * GPRN_TILE * ld + (int64_t)tj * GPRN_TILE;
            t.a_off = a_mode == 0 ? (int64_t)ti * GPRN_TILE * ld : (int64_t)ti * GPRN_TILE;
            t.b_off = b_mode == 0 ? (int64_t)tj * GPRN_TILE * ld : (int64_t)tj * GPRN_TILE;
            t.klen = K;
            t.c_buf = 2; t.a_buf = 0; t.b_buf = 1;
            t.modes = tile_modes(c_mode & 3, a_mode, b_mode);
            tasks.push_back(t);
        }
    TileTask* d_t = nullptr;
    double** d_p = nullptr;
    TRY(dev_alloc(c, &d_t, tasks.size()));
    TRY(dev_alloc(c, &d_p, GPRN_NBUF));
    double* hp[GPRN_NBUF] = {c->d_test[0], c->d_test[1], c->d_test[2], nullptr};
    HIP_TRY(c, hipMemcpy(d_t, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_p, hp, sizeof(hp), hipMemcpyHostToDevice));
    int rc = launch_tiles(c, d_t, tasks.size(), d_p, 1, ld, GPRN_T_UPDATE, nullptr, (c_mode >> 4) & 3);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e == hipSuccess)
        e = hipMemcpy(hc.data(), c->d_test[2], nn * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d_t); tab_forget(c, d_p); hipFree(d_p);
    if (rc) return rc;
    HIP_TRY(c, e);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) C[(size_t)m * N + n] = hc[(size_t)m * ld + n];
    return GPRN_OK;
}

// Rate of the tile contraction on an M x N x K product C -= A.B^T of random data already on the device (diagnostic):
// how = 0 / 1: one launch of the tile kernel, 64 x 64 / 128 x 128 workgroups.  ms: average of `reps` runs.
extern "C" int gprn_test_gemm_rate(gprn_ctx* c, int M, int N, int K, int how, int reps, double* ms)
{
    DeviceLock lock_(c);
    if (!c || M <= 0 || N <= 0 || K <= 0 || M % GPRN_TILE || N % GPRN_TILE || K % GPRN_KC || reps < 1 || !ms || how < 0 || how > 1)
        return bad(c, "test_gemm_rate: bad argument");
    const int ld = std::max(std::max(M, N), K);
    TRY(test_setup(c, ld, 3, 1));
    const size_t nn = (size_t)ld * ld;
    {
        std::vector<double> h(nn);
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < nn; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        for (int b = 0; b < 3; ++b) HIP_TRY(c, hipMemcpy(c->d_test[b], h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    }
    std::vector<TileTask> tasks;
    for (int ti = 0; ti < M / GPRN_TILE; ++ti)
        for (int tj = 0; tj < N / GPRN_TILE; ++tj)
            tasks.push_back(TileTask{(int64_t)ti * GPRN_TILE * ld + (int64_t)tj * GPRN_TILE, (int64_t)ti * GPRN_TILE * ld,
                                     (int64_t)tj * GPRN_TILE * ld, K, 2, 0, 1, tile_modes(CM_SUB, 0, 0)});
    TileTask* d_t = nullptr;
    double** d_p = nullptr;
    TRY(dev_alloc(c, &d_t, tasks.size()));
    TRY(dev_alloc(c, &d_p, GPRN_NBUF));
    double* hp[GPRN_NBUF] = {c->d_test[0], c->d_test[1], c->d_test[2], nullptr};
    HIP_TRY(c, hipMemcpy(d_t, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_p, hp, sizeof(hp), hipMemcpyHostToDevice));
    int rc = GPRN_OK;
    float t = 0.f;
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float total = 0.f;
        for (int r = 0; r < reps + 1 && !rc; ++r) {
            hipEventRecord(e0, c->stream);
            rc = launch_tiles(c, d_t, tasks.size(), d_p, 1, ld, GPRN_T_UPDATE, nullptr, how == 0 ? TS_64x64 : TS_128x128);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            float tt = 0.f;
            hipEventElapsedTime(&tt, e0, e1);
            if (r) total += tt;
        }
        t = total / reps;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipStreamSynchronize(c->stream);
    hipFree(d_t); tab_forget(c, d_p); hipFree(d_p);
    *ms = t;
    return rc;
}

// Time (ms per pass, average of `reps`) of the set-up's covariance fills -- every latent GP's kernel as last given by
// gprn_set_kernel into its own K, launch behind launch -- inside ONE pair of events: the rate the kernels run at.
// (The profiler's 'fill' family brackets every launch with events of its own: that figure includes the gaps between
// launches and varies with the box.)
extern "C" int gprn_test_fill_rate(gprn_ctx* c, int reps, double* ms)
{
    DeviceLock lock_(c);
    if (!c || !c->N || reps < 1 || !ms) return bad(c, "test_fill_rate: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(build_tables(c));
    std::vector<int> gps(c->loc_nodes);
    gps.insert(gps.end(), c->loc_weights.begin(), c->loc_weights.end());
    for (int g : gps)
        if (!c->kspec[g].set || c->kspec[g].uploaded) return bad(c, "test_fill_rate: every local latent GP needs a device kernel");
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    int rc = GPRN_OK;
    for (int g : gps) if (!rc) rc = launch_fill(c, c->kspec[g], c->K[g]);          // warm
    hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps && !rc; ++r)
        for (int g : gps) if (!rc) rc = launch_fill(c, c->kspec[g], c->K[g]);
    hipEventRecord(e1, c->stream);
    hipEventSynchronize(e1);
    float t = 0.f;
    hipEventElapsedTime(&t, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms = t / reps;
    return rc;
}

// run the library's own factorisation on caller matrices: temporarily a tiny "problem"
static int test_factor_impl(gprn_ctx* c, int n, int batch, const double* A, double* L,
                            double* Linv, bool lauum, double* lauum_out);

static int test_factor_common(gprn_ctx* c, int n, int batch, const double* A, double* L,
                              double* Linv, bool lauum, double* lauum_out)
{
    if (!c || n <= 0 || n % GPRN_TILE || batch <= 0 || !A) return bad(c, "test_factor: bad argument");
    return with_event_fallback(c, "test_factor", [&](bool) {
        return test_factor_impl(c, n, batch, A, L, Linv, lauum, lauum_out); });
}

static int test_factor_impl(gprn_ctx* c, int n, int batch, const double* A, double* L,
                            double* Linv, bool lauum, double* lauum_out)
{
    TRY(test_setup(c, n, 2, batch));
    const size_t nn = (size_t)n * n;
    HIP_TRY(c, hipMemcpy(c->d_test[0], A, nn * batch * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_test[1], 0, nn * batch * sizeof(double)));
    // borrow the context's factorisation state
    const int sN = c->N, sld = c->ld, sT = c->T, stT = c->tasks_T;
    double** sptrs = c->d_ptrs;
    int* sinfo = c->d_info_cur;
    c->N = n; c->ld = n; c->T = n / GPRN_TILE;
    double** d_p = nullptr;
    int* d_i = nullptr;
    int rc = dev_alloc(c, &d_p, (size_t)batch * GPRN_NBUF);
    if (!rc) rc = dev_alloc(c, &d_i, batch);
    std::vector<double*> hp((size_t)batch * GPRN_NBUF, nullptr);
    for (int b = 0; b < batch; ++b) {
        hp[(size_t)b * GPRN_NBUF + BUF_B] = c->d_test[0] + b * nn;
        hp[(size_t)b * GPRN_NBUF + BUF_X] = c->d_test[1] + b * nn;
    }
    hipError_t e = hipSuccess;
    int info0 = 0;
    if (!rc) {
        e = hipMemcpy(d_p, hp.data(), hp.size() * sizeof(double*), hipMemcpyHostToDevice);
        if (e == hipSuccess) tab_note(c, d_p, hp.data(), hp.size());
        if (e == hipSuccess) e = hipMemset(d_i, 0, batch * sizeof(int));
        c->d_ptrs = d_p; c->d_info_cur = d_i;
        c->tasks_T = -1;                       // force a task rebuild for this n
        if (e == hipSuccess) rc = lauum ? GPRN_OK : factor_invert(c, batch);
        if (lauum && e == hipSuccess) {
            // X := A (lower), out -> BUF_B
            e = hipMemcpy(c->d_test[1], A, nn * sizeof(double), hipMemcpyHostToDevice);
            if (e == hipSuccess) rc = lauum_lower(c, 1);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && !rc) rc = factor_check_waits(c);
        if (e == hipSuccess && !rc) {
            if (lauum) e = hipMemcpy(lauum_out, c->d_test[0], nn * sizeof(double), hipMemcpyDeviceToHost);
            else {
                e = hipMemcpy(L, c->d_test[0], nn * batch * sizeof(double), hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(Linv, c->d_test[1], nn * batch * sizeof(double), hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(&info0, d_i, sizeof(int), hipMemcpyDeviceToHost);
                for (int b = 0; b < batch; ++b)           // the upper triangle still holds A
                    for (int m = 0; m < n; ++m)
                        for (int k2 = m + 1; k2 < n; ++k2) L[b * nn + (size_t)m * n + k2] = 0.0;
            }
        }
    }
    if (d_p) { tab_forget(c, d_p); hipFree(d_p); }
    if (d_i) hipFree(d_i);
    c->N = sN; c->ld = sld; c->T = sT; c->d_ptrs = sptrs; c->d_info_cur = sinfo;
    c->tasks_T = -1;                           // the problem's own lists are rebuilt on demand
    (void)stT;
    if (rc) return rc;
    HIP_TRY(c, e);
    return info0;
}

extern "C" int gprn_test_factor_invert(gprn_ctx* c, int n, int batch, const double* A, double* L, double* Linv)
{
    DeviceLock lock_(c);
    if (!L || !Linv) return bad(c, "test_factor_invert: bad argument");
    return test_factor_common(c, n, batch, A, L, Linv, false, nullptr);
}

extern "C" int gprn_test_lauum(gprn_ctx* c, int n, const double* X, double* out)
{
    DeviceLock lock_(c);
    if (!out) return bad(c, "test_lauum: bad argument");
    return test_factor_common(c, n, 1, X, nullptr, nullptr, true, out);
}
