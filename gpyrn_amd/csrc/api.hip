// C ABI of libgprn_hip.so (include/gprn_hip.h): context, setup, the sweep loop.
#include "api_internal.h"

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

// ------------------------------------------------------------------ the collective watchdog
// A collective has no time-out of its own: a rank that dies inside one (or never enters it) leaves the others waiting --
// RCCL kernels spinning on the device, the host in the stream synchronisation behind them -- until somebody kills the job,
// holding their GPUs meanwhile.  Every entry point that issues collectives on a context with a communicator therefore
// runs under a watch: a detached thread (one per process) looks at the open watches every 200 ms, and when one has been
// open longer than the context's budget (option "comm_budget_s", default 600 s; GPRN_COMM_BUDGET_S) it says which entry
// point, which collective was enqueued last and which rank, and ends the process with a non-zero status (_exit: no
// restart, no re-exec, no unwinding through a runtime that is blocked).  The launcher then sees a failed rank and stops
// the others (bench.py's self-launcher, torchrun).

static std::mutex g_watch_mu;
static std::vector<WatchEntry*> g_watch;
static bool g_watch_thread = false;


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

int ensure_gp_storage(gprn_ctx* c, int g)
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
int comm_broadcast(gprn_ctx* c, double* buf, size_t n, int root)
{
    watch_note(c, "row broadcast");
    if (c->shm) return shm_broadcast(c, buf, n, root);
    NCCL_TRY(c, g_rccl.Broadcast(buf, buf, n, ncclDouble, root, (ncclComm_t)c->comm, c->stream));
    return GPRN_OK;
}

int comm_allreduce(gprn_ctx* c, double* buf, size_t n, bool is_max)
{
    watch_note(c, is_max ? "max all-reduce (agreement / barrier)" : "sum all-reduce (per-GP scalars)");
    if (c->shm) return shm_allreduce(c, buf, n, is_max);
    NCCL_TRY(c, g_rccl.AllReduce(buf, buf, n, ncclDouble, is_max ? ncclMax : ncclSum, (ncclComm_t)c->comm, c->stream));
    return GPRN_OK;
}

bool comm_active(const gprn_ctx* c) { return c->comm || c->shm; }

// several broadcasts as ONE grouped RCCL call (nothing to do on the shm transport)
int comm_group(gprn_ctx* c, bool begin)
{
    if (!c->comm) return GPRN_OK;
    if ((begin ? g_rccl.GroupStart() : g_rccl.GroupEnd()) != ncclSuccess) { c->err = begin ? "ncclGroupStart" : "ncclGroupEnd"; return GPRN_E_COMM; }
    return GPRN_OK;
}

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
int exchange_rows(gprn_ctx* c, bool weights)
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

int reduce_scalars(gprn_ctx* c)
{
    if (!comm_active(c)) return GPRN_OK;
    const size_t n = 3 * (size_t)c->G + (size_t)c->q * c->q;
    return comm_allreduce(c, c->d_scal, n, false);
}

// ------------------------------------------------------------------ tables
int upload_table(gprn_ctx* c, double** d_tab, const std::vector<double*>& rows)
{
    HIP_TRY(c, hipMemcpyAsync(d_tab, rows.data(), rows.size() * sizeof(double*),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); watch_progress(c);
    tab_note(c, d_tab, rows.data(), rows.size());
    return GPRN_OK;
}

int build_tables(gprn_ctx* c)
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

int check_info(gprn_ctx* c, const int* d_info, const std::vector<int>& gps, int* first)
{
    std::vector<int> h(gps.size());
    if (gps.empty()) return GPRN_OK;
    HIP_TRY(c, hipMemcpy(h.data(), d_info, gps.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < gps.size(); ++s)
        if (h[s] > 0 && *first == 0) { *first = h[s]; c->info_gp = gps[s]; }
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

