// Shared by the translation units of the C ABI (api.hip: context, problem, options, watchdog, communicator, tables, read-back;
// api_sweep.hip: the set-up, the sweep, the ELBOcalc loop; api_more.hip: prediction, kernel matrices and prior draws, gradients,
// the ELBO's terms on their own, diagnostics).  Round 6 split of a 2 700-line api.hip along its section banners; the 42
// exported symbols are unchanged (tests/test_abi.py).
#pragma once
#include "gprn_internal.h"
#include "vecops.h"

#include <atomic>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>

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

static inline int bad(gprn_ctx* c, const char* msg) { if (c) c->err = msg; return GPRN_E_ARG; }

// Runs `body` (an entry point that factorises); when one of its in-kernel dependency waits gave up
// (GPRN_E_WAIT_TIMEOUT: a serialising tool, a starved device, ...), the context is latched to the event
// schedule and the body runs once more -- `body` must restore what it changed before it starts over.
// On a sharded context the verdict is shared first (one max-reduce of the "timed out" word per call): the body
// issues collectives, so either every rank runs it again or none does -- a rank re-running on its own would issue
// its broadcasts and its all-reduce a second time while the others have moved on.
int comm_allreduce(gprn_ctx* c, double* buf, size_t n, bool is_max);
int comm_broadcast(gprn_ctx* c, double* buf, size_t n, int root);
bool comm_active(const gprn_ctx* c);
// (the collective watchdog, below: EVERY stream synchronisation of this file that returns tells it the device has made
// progress -- the budget bounds a stall, not the length of a call: ADVICE r5, gprn_elbocalc at N = 16384 runs for minutes)
static inline void watch_progress(gprn_ctx* c);

static inline int agree_on_timeout(gprn_ctx* c, int rc, bool* any)
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
static inline int agree_to_start(gprn_ctx* c, int local_rc, const char* what)
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


// ---- the collective watchdog (api.hip): what the entry points need of it
struct WatchEntry {
    std::atomic<long long> since_ms{0};        // 0: nothing open
    std::atomic<const char*> what{nullptr};    // the entry point
    std::atomic<const char*> last{nullptr};    // the collective enqueued last
    std::atomic<int> budget_s{600}, rank{0}, world{1};
};
static inline long long now_ms()
{
    timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (long long)t.tv_sec * 1000 + t.tv_nsec / 1000000;
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


// ---- defined in api.hip
int comm_group(gprn_ctx* c, bool begin);     // ncclGroupStart / End around several broadcasts
int exchange_rows(gprn_ctx* c, bool weights);
int reduce_scalars(gprn_ctx* c);
int upload_table(gprn_ctx* c, double** d_tab, const std::vector<double*>& rows);
int build_tables(gprn_ctx* c);
int check_info(gprn_ctx* c, const int* d_info, const std::vector<int>& gps, int* first);
int ensure_gp_storage(gprn_ctx* c, int g);
// ---- defined in api_more.hip (diagnostics): a scratch "problem" of `batch` ld x ld matrices
int test_setup(gprn_ctx* c, int ld, int nbuf_needed, int batch);
