// Many independent evaluations of ONE problem of more than one tile, side by side (gprn_elbocalc_batch for N > 128).
//
// The reference's realistic callers -- scipy's simplex under inference.optimize, emcee's walkers under inference.mcmc
// (meanfield.py:1095-1152, 1222-1260) -- ask for nELBO at one parameter vector after the other, on problems of a few
// hundred points (its only real dataset: gpyrn/datasets/Solar_observations.txt, 497 rows).  At N = 512 one evaluation is
// a chain of ~400 launches that keeps a handful of the device's 256 CUs busy: 3.3 ms, all of it launch latency.  The
// evaluations are independent, so B of them go through the SAME launch sequence with the batch dimension of every launch
// = evaluations x latent GPs of the phase (factor_invert: the tile kernels' grid y): the chain of a phase is walked once
// per batch instead of once per evaluation, and every launch has B times the workgroups.
//
// A WORKER context (a gprn_ctx of its own on the parent's device and streams) holds a chunk of evaluations: per
// evaluation and latent GP the prior matrix K, chol(K)^-1, the sweep's workspaces B and X, K_j^-1 for nodes j >= 1
// (quirk Q1), and per evaluation the state, y - mean, the variances, the per-GP scalars -- the kernels of vecops.hip find
// an evaluation's copy through gprn_ctx::ev (slot -> evaluation, strides).  The parent's own state and factors are not
// touched.  Per sweep: node phase (phase_core, api_sweep.hip), the Q1 products, weight phase, the prior terms, the ELBO of
// every evaluation still running, ONE read-back (4 doubles per evaluation + the pivot verdicts); the stop rule of
// meanfield.py:640-643 is applied per evaluation on the host, and an evaluation that has stopped leaves the tables of the
// next sweep (its slots are compacted away: the launches shrink with the number of evaluations still running).
// Lists longer than the memory budget (option "batch_mem_mb") run chunk by chunk.
#include "gprn_internal.h"
#include "vecops.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <vector>

#define MB_TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)
#define MID_LEAD 4     // sweeps enqueued before the host looks at their results (the stop rule cannot fire before trip 4)

struct MidBatch {
    gprn_ctx* w = nullptr;            // the worker context
    int cap = 0;                      // evaluations the slabs hold
    int N = 0, p = 0, q = 0, G = 0, ld = 0;
    double *K = nullptr, *KL = nullptr, *Bw = nullptr, *Xw = nullptr;   // [cap][G][ld * ld]
    double *Kinv = nullptr;           // [cap][q - 1][ld * ld]: K_j^-1 (lower), j = 1 .. q - 1
    double *q1_scratch = nullptr;     // [cap q (q - 1) / 2][ld]
    void* programs = nullptr;         // [cap][G] fill programs
    // device tables, one allocation: pointers first, then ints
    double** d_ptr_block = nullptr;   // kptr [cap G] | kptr2 [cap G] | tab_setup [cap G][4] | tab_kinv [cap (q-1)][4] | tab_node [cap q][4] | tab_weight [cap qp][4]
    int* d_int_block = nullptr;       // gp_setup [cap G] | ev_setup [cap G] | gp_node, ev_node [cap q] | gp_weight, ev_weight [cap qp] | evals [cap]
    size_t n_ptr = 0, n_int = 0;
    char *pin_in = nullptr, *pin_out = nullptr, *pin_tab = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0, pin_tab_bytes = 0;
    // offsets into the blocks
    size_t o_kptr = 0, o_kptr2 = 0, o_setup = 0, o_kinv = 0, o_node = 0, o_weight = 0;
    size_t i_gp_setup = 0, i_ev_setup = 0, i_gp_node = 0, i_ev_node = 0, i_gp_weight = 0, i_ev_weight = 0, i_evals = 0;
};

static void mid_free_slabs(MidBatch* m)
{
    void* dev[] = {m->K, m->KL, m->Bw, m->Xw, m->Kinv, m->q1_scratch, m->programs, m->d_ptr_block, m->d_int_block};
    for (void* ptr : dev) if (ptr) hipFree(ptr);
    m->K = m->KL = m->Bw = m->Xw = m->Kinv = m->q1_scratch = nullptr;
    m->programs = nullptr; m->d_ptr_block = nullptr; m->d_int_block = nullptr;
    if (m->pin_in) hipHostFree(m->pin_in);
    if (m->pin_out) hipHostFree(m->pin_out);
    if (m->pin_tab) hipHostFree(m->pin_tab);
    m->pin_in = m->pin_out = m->pin_tab = nullptr;
    m->cap = 0;
}

void mid_batch_free(gprn_ctx* c)
{
    MidBatch* m = (MidBatch*)c->mid_batch;
    if (!m) return;
    mid_free_slabs(m);
    if (m->w) gprn_destroy(m->w);
    delete m;
    c->mid_batch = nullptr;
}

template <typename TT>
static int mb_alloc(gprn_ctx* c, TT** ptr, size_t count)
{
    *ptr = nullptr;
    if (hipMalloc((void**)ptr, std::max<size_t>(count, 1) * sizeof(TT)) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "hipMalloc (evaluation batch, N > 128)";
        return GPRN_E_NOMEM;
    }
    return GPRN_OK;
}

// device bytes one evaluation of this problem takes in a chunk
static size_t mid_bytes_per_eval(const gprn_ctx* c)
{
    const size_t nn = (size_t)c->ld * c->ld, G = c->G, ld = c->ld;
    const size_t d = (size_t)(c->p + 1) * c->q * c->N;
    size_t dbl = (4 * G + (size_t)(c->q - 1)) * nn             // K, KL, B, X, K_j^-1
               + G * ld * (7 + 2 * (size_t)c->T + 2)           // per-slot vectors, partial column sums, finalising terms
               + (size_t)c->q * (c->q - 1) / 2 * ld + 2 * d + 2 * (size_t)c->p * c->N + 64;
    return dbl * sizeof(double);
}

// The worker context and the slabs for `want` evaluations (never more than the budget allows; at least one).
static int mid_ensure(gprn_ctx* c, int want, int* cap_out)
{
    MidBatch* m = (MidBatch*)c->mid_batch;
    const size_t per = mid_bytes_per_eval(c);
    // (a launch's grid y is the number of slots of a phase: cap x G stays far below its 65 535 limit)
    const int fit = (int)std::max<size_t>(1, std::min<size_t>(batch_budget_bytes(c) / per, (size_t)32768 / c->G));
    want = std::min(want, fit);
    const bool same = m && m->N == c->N && m->p == c->p && m->q == c->q && m->ld == c->ld;
    if (same && m->cap >= want && m->cap <= fit) { *cap_out = m->cap; return GPRN_OK; }
    if (m && !same) { mid_batch_free(c); m = nullptr; }
    if (!m) { m = new MidBatch(); c->mid_batch = m; }
    mid_free_slabs(m);
    const int N = c->N, p = c->p, q = c->q, G = c->G, ld = c->ld, T = c->T;
    m->N = N; m->p = p; m->q = q; m->G = G; m->ld = ld;
    // ---- the worker: the parent's problem, `cap` evaluations' worth of state and per-slot vectors
    if (!m->w) {
        const int rc = gprn_create(&m->w, c->device);
        if (rc) { c->err = "evaluation batch: cannot create the worker context"; return rc; }
    }
    gprn_ctx* w = m->w;
    const int cap = want;
    *cap_out = cap;                                       // (what was tried, for a caller that halves after GPRN_E_NOMEM)
    const size_t nn = (size_t)ld * ld, d = (size_t)(p + 1) * q * N, pn = (size_t)p * N, nscal = 3 * (size_t)G + (size_t)q * q;
    const size_t nslot = (size_t)cap * G;
    {
        // (free_problem's fields, sized for cap evaluations; everything a phase's launchers read from the context)
        void* old[] = {w->d_time, w->d_yraw, w->d_mu, w->d_var, w->d_yres, w->d_variance, w->d_logdetK, w->d_scal_base, w->d_elbo_part,
                       w->d_out, w->d_d, w->d_s, w->d_pred, w->d_z, w->d_u, w->d_cs, w->d_ct, w->d_part, w->d_info, w->d_fin_terms,
                       w->d_fin_tickets};
        for (void* ptr : old) if (ptr) hipFree(ptr);
        w->d_time = w->d_yraw = w->d_mu = w->d_var = w->d_yres = w->d_variance = w->d_logdetK = w->d_scal_base = nullptr;
        w->d_elbo_part = w->d_out = w->d_d = w->d_s = w->d_pred = w->d_z = w->d_u = w->d_cs = w->d_ct = w->d_part = nullptr;
        w->d_info = nullptr; w->d_fin_terms = nullptr; w->d_fin_tickets = nullptr;
    }
    w->N = N; w->p = p; w->q = q; w->G = G; w->ld = ld; w->T = T;
    w->h_yerr2 = c->h_yerr2;
    w->world = 1; w->rank = 0;
    w->owner.assign(G, 0);
    w->nslot = (int)nslot;
    w->out_cap = cap;
    MB_TRY(mb_alloc(c, &w->d_time, (size_t)N));
    MB_TRY(mb_alloc(c, &w->d_yraw, pn));
    MB_TRY(mb_alloc(c, &w->d_mu, (size_t)cap * d));
    MB_TRY(mb_alloc(c, &w->d_var, (size_t)cap * d));
    MB_TRY(mb_alloc(c, &w->d_yres, (size_t)cap * pn));
    MB_TRY(mb_alloc(c, &w->d_variance, (size_t)cap * pn));
    MB_TRY(mb_alloc(c, &w->d_logdetK, (size_t)cap * G));
    MB_TRY(mb_alloc(c, &w->d_scal_base, (size_t)cap * nscal));
    MB_TRY(mb_alloc(c, &w->d_elbo_part, (size_t)cap * GPRN_ELBO_PART_DOUBLES));
    MB_TRY(mb_alloc(c, &w->d_out, (size_t)MID_LEAD * cap * 4));      // (one block of results per sweep enqueued ahead)
    MB_TRY(mb_alloc(c, &w->d_d, nslot * ld)); MB_TRY(mb_alloc(c, &w->d_s, nslot * ld)); MB_TRY(mb_alloc(c, &w->d_pred, nslot * ld));
    MB_TRY(mb_alloc(c, &w->d_z, nslot * ld)); MB_TRY(mb_alloc(c, &w->d_u, nslot * ld)); MB_TRY(mb_alloc(c, &w->d_cs, nslot * ld));
    MB_TRY(mb_alloc(c, &w->d_ct, nslot * ld));
    MB_TRY(mb_alloc(c, &w->d_part, nslot * T * 2 * ld));
    MB_TRY(mb_alloc(c, &w->d_info, 3 * nslot));
    HIP_TRY(c, hipMemcpy(w->d_time, c->d_time, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice));
    HIP_TRY(c, hipMemcpy(w->d_yraw, c->d_yraw, pn * sizeof(double), hipMemcpyDeviceToDevice));
    HIP_TRY(c, hipMemset(w->d_scal_base, 0, (size_t)cap * nscal * sizeof(double)));
    HIP_TRY(c, hipMemset(w->d_info, 0, 3 * nslot * sizeof(int)));
    w->d_scal = w->d_scal_base;
    w->d_logdetB = w->d_scal; w->d_trBinv = w->d_scal + G; w->d_muKmu = w->d_scal + 2 * (size_t)G; w->d_q1 = w->d_scal + 3 * (size_t)G;
    w->ev = EvalMap{nullptr, d, pn, nscal, (size_t)G};
    w->have_yres = w->have_jit = w->have_muvar = true;
    // ---- the slabs
    MB_TRY(mb_alloc(c, &m->K, (size_t)cap * G * nn));
    MB_TRY(mb_alloc(c, &m->KL, (size_t)cap * G * nn));
    MB_TRY(mb_alloc(c, &m->Bw, (size_t)cap * G * nn));
    MB_TRY(mb_alloc(c, &m->Xw, (size_t)cap * G * nn));
    if (q > 1) {
        MB_TRY(mb_alloc(c, &m->Kinv, (size_t)cap * (q - 1) * nn));
        MB_TRY(mb_alloc(c, &m->q1_scratch, (size_t)cap * (q * (q - 1) / 2) * ld));
    }
    if (hipMalloc(&m->programs, (size_t)cap * G * fill_program_bytes()) != hipSuccess) { c->err = "hipMalloc (fill programs)"; return GPRN_E_NOMEM; }
    // ---- tables
    const size_t qp = (size_t)q * p;
    m->o_kptr = 0;
    m->o_kptr2 = m->o_kptr + nslot;
    m->o_setup = m->o_kptr2 + nslot;
    m->o_kinv = m->o_setup + nslot * GPRN_NBUF;
    m->o_node = m->o_kinv + (size_t)cap * (q - 1) * GPRN_NBUF;
    m->o_weight = m->o_node + (size_t)cap * q * GPRN_NBUF;
    m->n_ptr = m->o_weight + (size_t)cap * qp * GPRN_NBUF;
    m->i_gp_setup = 0; m->i_ev_setup = nslot;
    m->i_gp_node = 2 * nslot; m->i_ev_node = m->i_gp_node + (size_t)cap * q;
    m->i_gp_weight = m->i_ev_node + (size_t)cap * q; m->i_ev_weight = m->i_gp_weight + (size_t)cap * qp;
    m->i_evals = m->i_ev_weight + (size_t)cap * qp;
    m->n_int = m->i_evals + cap;
    MB_TRY(mb_alloc(c, &m->d_ptr_block, m->n_ptr));
    MB_TRY(mb_alloc(c, &m->d_int_block, m->n_int));
    m->pin_tab_bytes = m->n_ptr * sizeof(double*) + m->n_int * sizeof(int);
    m->pin_in_bytes = (size_t)cap * G * fill_program_bytes() + (2 * (size_t)cap * pn + 2 * (size_t)cap * d) * sizeof(double);
    m->pin_out_bytes = (size_t)MID_LEAD * cap * 4 * sizeof(double) + 3 * nslot * sizeof(int) + 2 * (size_t)cap * d * sizeof(double) + 64;
    HIP_TRY(c, hipHostMalloc((void**)&m->pin_tab, m->pin_tab_bytes, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&m->pin_in, m->pin_in_bytes, hipHostMallocDefault));
    HIP_TRY(c, hipHostMalloc((void**)&m->pin_out, m->pin_out_bytes, hipHostMallocDefault));
    // the tables of the set-up never change: slot = evaluation * G + latent GP
    {
        double** hp = (double**)m->pin_tab;
        int* hi = (int*)(m->pin_tab + m->n_ptr * sizeof(double*));
        for (int b = 0; b < cap; ++b)
            for (int g = 0; g < G; ++g) {
                const size_t s = (size_t)b * G + g;
                hp[m->o_kptr + s] = m->K + s * nn;
                hp[m->o_kptr2 + s] = m->Bw + s * nn;          // (the set-up factors a copy of K in place: the fill writes both)
                double** row = hp + m->o_setup + s * GPRN_NBUF;
                row[BUF_B] = m->Bw + s * nn; row[BUF_X] = m->KL + s * nn; row[BUF_K] = m->K + s * nn; row[BUF_KLINV] = m->KL + s * nn;
                hi[m->i_gp_setup + s] = g;
                hi[m->i_ev_setup + s] = b;
            }
        for (int b = 0; b < cap; ++b)
            for (int j = 1; j < q; ++j) {                      // lower(K_j^-1) = lower(X^T X), X = chol(K_j)^-1
                const size_t s = (size_t)b * (q - 1) + (j - 1);
                double** row = hp + m->o_kinv + s * GPRN_NBUF;
                row[BUF_B] = m->Kinv + s * nn; row[BUF_X] = m->KL + ((size_t)b * G + j) * nn;
                row[BUF_K] = nullptr; row[BUF_KLINV] = nullptr;
            }
        HIP_TRY(c, hipMemcpy(m->d_ptr_block, hp, (m->o_node) * sizeof(double*), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(m->d_int_block, hi, (m->i_gp_node) * sizeof(int), hipMemcpyHostToDevice));
    }
    m->cap = cap;
    *cap_out = cap;
    return GPRN_OK;
}

// The tables of a sweep over the evaluations `act` (positions in the chunk): node slots node-major (slot = j nA + a: the
// first (q - 1) nA slots are the nodes whose B^-1 quirk Q1 needs), weight slots likewise; one copy.
static int mid_upload_active(gprn_ctx* c, MidBatch* m, const std::vector<int>& act)
{
    gprn_ctx* w = m->w;
    const int nA = (int)act.size(), q = m->q, p = m->p, G = m->G;
    const size_t nn = (size_t)m->ld * m->ld, qp = (size_t)q * p;
    double** hp = (double**)m->pin_tab;
    int* hi = (int*)(m->pin_tab + m->n_ptr * sizeof(double*));
    auto put = [&](double** row, int b, int g) {
        const size_t s = (size_t)b * G + g;
        row[BUF_B] = m->Bw + s * nn; row[BUF_X] = m->Xw + s * nn; row[BUF_K] = m->K + s * nn; row[BUF_KLINV] = m->KL + s * nn;
    };
    for (int j = 0; j < q; ++j)
        for (int a = 0; a < nA; ++a) {
            const size_t s = (size_t)j * nA + a;
            put(hp + m->o_node + s * GPRN_NBUF, act[a], j);
            hi[m->i_gp_node + s] = j;
            hi[m->i_ev_node + s] = act[a];
        }
    for (size_t kk = 0; kk < qp; ++kk)
        for (int a = 0; a < nA; ++a) {
            const size_t s = kk * nA + a;
            put(hp + m->o_weight + s * GPRN_NBUF, act[a], q + (int)kk);
            hi[m->i_gp_weight + s] = q + (int)kk;
            hi[m->i_ev_weight + s] = act[a];
        }
    for (int a = 0; a < nA; ++a) hi[m->i_evals + a] = act[a];
    // (two pieces each: the node and weight tables lie side by side in both blocks)
    HIP_TRY(c, hipMemcpyAsync(m->d_ptr_block + m->o_node, hp + m->o_node, (m->n_ptr - m->o_node) * sizeof(double*),
                              hipMemcpyHostToDevice, w->stream));
    HIP_TRY(c, hipMemcpyAsync(m->d_int_block + m->i_gp_node, hi + m->i_gp_node, (m->n_int - m->i_gp_node) * sizeof(int),
                              hipMemcpyHostToDevice, w->stream));
    return GPRN_OK;
}

static int mid_phase(gprn_ctx* w, MidBatch* m, bool weights, int nA)
{
    const int per = weights ? m->q * m->p : m->q, ns = per * nA;
    w->d_ptrs = m->d_ptr_block + (weights ? m->o_weight : m->o_node);
    w->ev.slot_eval = m->d_int_block + (weights ? m->i_ev_weight : m->i_ev_node);
    w->slot0 = weights ? nA * m->q : 0;
    w->d_info_cur = w->d_info + (weights ? 2 : 1) * (size_t)w->nslot;
    return phase_core(w, weights, m->d_int_block + (weights ? m->i_gp_weight : m->i_gp_node), ns);
}

// m^T K^-1 m = |L_K^-1 m|^2 per latent GP of the phase, m the state row as it lies in memory (quirk Q2)
static int mid_prior_term(gprn_ctx* w, MidBatch* m, bool weights, int nA, hipStream_t st)
{
    const int per = weights ? m->q * m->p : m->q, ns = per * nA;
    const int* slotgp = m->d_int_block + (weights ? m->i_gp_weight : m->i_gp_node);
    w->d_ptrs = m->d_ptr_block + (weights ? m->o_weight : m->o_node);
    w->ev.slot_eval = m->d_int_block + (weights ? m->i_ev_weight : m->i_ev_node);
    w->slot0 = weights ? nA * m->q : 0;
    double* a = w->d_u + (size_t)w->slot0 * w->ld;
    MB_TRY(vec_lower_matvec(w, BUF_KLINV, w->d_mu, w->N, 1, slotgp, ns, a, st));
    return vec_dot_self(w, slotgp, ns, a, w->d_muKmu, st);
}

// One sweep (meanfield.py:651-710) of the evaluations in the active tables; out4 of each lands at out4 + 4 * evaluation.
// The pivot verdicts of its phases are only raised (rows 1, 2 of d_info: the caller clears them).
static int mid_sweep(gprn_ctx* w, MidBatch* m, int nA, double* out4)
{
    w->chain_started = nullptr;                          // (nothing left over from a sweep that broke off)
    MB_TRY(mid_phase(w, m, false, nA));
    // What reads the node phase's results and nothing of the weight phase's runs BESIDE that phase on the bulk stream, handed to
    // its factorisation (behind the first diagonal block, as run_phase does it for one evaluation) and joined before the
    // ELBO assembly: the nodes' prior term m^T K^-1 m, and quirk Q1 (:1039-1041) -- lower(B_k^-1) = lower(X^T X) of every
    // node but the last into its B buffer (L is not needed any more: log det B is taken), then <K_j^-1, Sigma_k> for j > k.
    HIP_TRY(w, hipEventRecord(w->ev_nodes, w->stream));
    w->chain_started = [w, m, nA]() -> int {
        double** const cur = w->d_ptrs;
        const int cur_slot0 = w->slot0;
        const int* const cur_ev = w->ev.slot_eval;
        HIP_TRY(w, hipStreamWaitEvent(w->stream2, w->ev_nodes, 0));
        int rc = mid_prior_term(w, m, false, nA, w->stream2);
        if (!rc && m->q > 1) {
            w->d_ptrs = m->d_ptr_block + m->o_node;
            rc = lauum_lower(w, (m->q - 1) * nA, w->stream2);
            if (!rc) rc = vec_q1_evals(w, m->d_int_block + m->i_ev_node, m->Kinv, nA, m->q1_scratch, w->stream2);
        }
        w->d_ptrs = cur; w->slot0 = cur_slot0; w->ev.slot_eval = cur_ev;
        if (rc) return rc;
        HIP_TRY(w, hipEventRecord(w->ev_q1, w->stream2));
        return GPRN_OK;
    };
    MB_TRY(mid_phase(w, m, true, nA));
    if (w->chain_started) {                              // (no factorisation took it along)
        std::function<int()> f;
        f.swap(w->chain_started);
        MB_TRY(f());
    }
    HIP_TRY(w, hipStreamWaitEvent(w->stream, w->ev_q1, 0));
    MB_TRY(mid_prior_term(w, m, true, nA, w->stream));
    return vec_elbo_evals(w, m->d_int_block + m->i_evals, nA, out4, w->d_scal_base, w->d_elbo_part);
}

struct MidIo {
    int n; const double *kparams; int n_kpar; const double *y_resid, *jitters, *mu, *var; int max_iter;
    double* elbo; int *iters, *conv, *info; double *mu_out, *var_out;
};

// One chunk of evaluations (n <= cap) from staging to results; restartable (everything it reads is the caller's).
static int mid_chunk(gprn_ctx* c, MidBatch* m, const MidIo& io)
{
    gprn_ctx* w = m->w;
    const int B = io.n, G = m->G, p = m->p, q = m->q, N = m->N;
    const size_t d = (size_t)(p + 1) * q * N, pn = (size_t)p * N, pb = fill_program_bytes();
    hipStream_t st = w->stream;
    // GPRN_BATCH_TIMERS=1 (probes): where the host's time of a chunk goes, on stderr
    static int timers_env = -1;
    if (timers_env < 0) { const char* e = getenv("GPRN_BATCH_TIMERS"); timers_env = e ? atoi(e) : 0; }
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_mark = t_begin;
    auto lap = [&]() { const auto now = std::chrono::steady_clock::now(); const double us = std::chrono::duration<double, std::micro>(now - t_mark).count(); t_mark = now; return us; };
    double us_stage = 0.0, us_setup = 0.0, us_enqueue = 0.0, us_wait = 0.0, us_host = 0.0;
    int n_sweeps = 0;
    // ---- inputs through the pinned buffer: programs | y - mean | variance | mu | var
    char* const pg_h = m->pin_in;
    double* const yres_h = (double*)(pg_h + (size_t)m->cap * G * pb);
    double* const var_h = yres_h + (size_t)m->cap * pn;
    double* const mu0_h = var_h + (size_t)m->cap * pn;
    double* const v0_h = mu0_h + (size_t)m->cap * d;
    for (int b = 0; b < B; ++b) {
        const double* kp = io.kparams + (size_t)b * io.n_kpar;
        for (int g = 0; g < G; ++g) {
            if (!fill_program_with(c->kspec[g], kp, pg_h + ((size_t)b * G + g) * pb)) {
                c->err = "elbocalc_batch: a kernel that is not an even function of t_i - t_j"; return GPRN_E_UNSUPPORTED;
            }
            kp += c->kspec[g].n_params;
        }
        for (int i = 0; i < p; ++i) {
            const double j2 = io.jitters[(size_t)b * p + i] * io.jitters[(size_t)b * p + i];
            for (int n = 0; n < N; ++n) var_h[(size_t)b * pn + (size_t)i * N + n] = j2 + c->h_yerr2[(size_t)i * N + n];
        }
    }
    memcpy(yres_h, io.y_resid, (size_t)B * pn * sizeof(double));
    memcpy(mu0_h, io.mu, (size_t)B * d * sizeof(double));
    memcpy(v0_h, io.var, (size_t)B * d * sizeof(double));
    us_stage = lap();
    HIP_TRY(c, hipMemcpyAsync(m->programs, pg_h, (size_t)B * G * pb, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(w->d_yres, yres_h, (size_t)B * pn * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(w->d_variance, var_h, (size_t)B * pn * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(w->d_mu, mu0_h, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(w->d_var, v0_h, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
    // ---- set-up (meanfield.py:619-622): every evaluation's G covariance matrices in one launch, chol(K) and its inverse for
    // all of them in one factorisation, log det K, and K_j^-1 = X^T X for the nodes quirk Q1 needs
    w->ev.slot_eval = m->d_int_block + m->i_ev_setup;
    MB_TRY(launch_fill_batch(w, m->programs, (double* const*)(m->d_ptr_block + m->o_kptr), B * G,
                             (double* const*)(m->d_ptr_block + m->o_kptr2)));
    HIP_TRY(c, hipMemsetAsync(w->d_info, 0, 3 * (size_t)w->nslot * sizeof(int), st));
    w->d_ptrs = m->d_ptr_block + m->o_setup;
    w->slot0 = 0;
    w->d_info_cur = w->d_info;
    MB_TRY(factor_invert(w, B * G, true));
    MB_TRY(vec_logdet(w, BUF_B, m->d_int_block + m->i_gp_setup, B * G, w->d_logdetK));
    if (q > 1) {
        w->d_ptrs = m->d_ptr_block + m->o_kinv;
        MB_TRY(lauum_lower(w, B * (q - 1)));
    }
    // ---- the loop of meanfield.py:626-649, per evaluation.  Quirk Q7: the first ELBOaux call (update discarded, ELBO kept
    // as elboArray[0]) and the loop's first trip are the same computation on the same input -- it runs once and its value
    // is entered twice (max_iter = 0: the sweep runs, the state the caller gave is what comes back).
    us_setup = lap();
    std::vector<int> act(B);
    for (int b = 0; b < B; ++b) { act[b] = b; io.elbo[b] = 0.0; io.iters[b] = 0; io.conv[b] = 0; io.info[b] = 0; }
    std::vector<double> last3((size_t)3 * B, 0.0);
    double* const out_h = (double*)m->pin_out;
    int* const info_h = (int*)(out_h + (size_t)MID_LEAD * m->cap * 4);
    bool tables_stale = true, first = true;
    while (!act.empty()) {
        const int nA = (int)act.size();
        if (tables_stale) { MB_TRY(mid_upload_active(c, m, act)); tables_stale = false; }
        // The stop rule cannot fire before trip 4 (:640), so the first trips -- min(4, max_iter) of them -- are enqueued
        // without looking at their results in between: one host round trip instead of four (80 us each: the read-back,
        // the rule, the next sweep's first launches), and the device goes from one sweep into the next.  A warm-started
        // evaluation -- nELBO's case -- usually stops right there.  Later trips go one by one: each may be an
        // evaluation's last, and its state must stay what that trip left.
        const int lead = first ? std::max(1, std::min(MID_LEAD, io.max_iter)) : 1;
        HIP_TRY(w, hipMemsetAsync(w->d_info + (size_t)w->nslot, 0, 2 * (size_t)w->nslot * sizeof(int), st));
        for (int sw = 0; sw < lead; ++sw) MB_TRY(mid_sweep(w, m, nA, w->d_out + (size_t)sw * m->cap * 4));
        HIP_TRY(c, hipMemcpyAsync(out_h, w->d_out, (size_t)lead * m->cap * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(info_h, w->d_info, 3 * (size_t)w->nslot * sizeof(int), hipMemcpyDeviceToHost, st));
        us_enqueue += lap();
        HIP_TRY(c, hipStreamSynchronize(st));
        us_wait += lap();
        n_sweeps += lead;
        MB_TRY(factor_check_waits(w));
        std::vector<int> next;
        next.reserve(nA);
        for (int a = 0; a < nA; ++a) {
            const int b = act[a];
            // pivot verdicts (raised, never lowered, by every sweep of the group): the set-up's (slot = b G + g) with the
            // first group, the phases' (node-major slots) always
            int bad = 0;
            if (first) for (int g = 0; g < G && !bad; ++g) bad = std::max(0, info_h[(size_t)b * G + g]);
            for (int j = 0; j < q && !bad; ++j) bad = std::max(0, info_h[(size_t)w->nslot + (size_t)j * nA + a]);
            for (int kk = 0; kk < q * p && !bad; ++kk) bad = std::max(0, info_h[2 * (size_t)w->nslot + (size_t)kk * nA + a]);
            double* l3 = &last3[(size_t)3 * b];
            bool go_on = true;
            for (int sw = 0; sw < lead && go_on; ++sw) {
                const double e = out_h[((size_t)sw * m->cap + b) * 4];
                if (bad || e != e) {
                    // a matrix that is not positive definite (jnp.linalg.cholesky: NaN from there on, no exception -- :71-89), or
                    // a state that has left the finite numbers: NaN stays NaN, so the loop would run to max_iter and return it
                    io.info[b] = bad;
                    io.elbo[b] = NAN;
                    io.iters[b] = io.max_iter;
                    go_on = false;
                    break;
                }
                io.elbo[b] = e;
                if (io.iters[b] == 0) {                     // the sweep that stands for ELBOaux call 0 and trip 1
                    l3[1] = e; l3[2] = e;
                    if (io.max_iter == 0) { go_on = false; break; }   // only the discarded sweep: done, state as given
                    io.iters[b] = 1;
                } else {
                    l3[0] = l3[1]; l3[1] = l3[2]; l3[2] = e;
                    io.iters[b] += 1;
                }
                // (inside a group neither can happen before its last sweep: lead <= min(4, max_iter))
                if (io.iters[b] > 3 && elbo_stop_rule(l3[0], l3[1], l3[2])) { io.conv[b] = 1; go_on = false; }
                else if (io.iters[b] >= io.max_iter) go_on = false;
            }
            if (go_on) next.push_back(b);
        }
        if (next.size() != act.size()) tables_stale = true;
        act.swap(next);
        first = false;
        us_host += lap();
    }
    if (io.mu_out && io.var_out) {
        double* const st_h = (double*)(((uintptr_t)(info_h + 3 * (size_t)w->nslot) + 63) & ~(uintptr_t)63);
        HIP_TRY(c, hipMemcpyAsync(st_h, w->d_mu, (size_t)B * d * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(st_h + (size_t)m->cap * d, w->d_var, (size_t)B * d * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        memcpy(io.mu_out, st_h, (size_t)B * d * sizeof(double));
        memcpy(io.var_out, st_h + (size_t)m->cap * d, (size_t)B * d * sizeof(double));
        if (io.max_iter == 0) {                          // (the one sweep's update is the discarded one)
            memcpy(io.mu_out, io.mu, (size_t)B * d * sizeof(double));
            memcpy(io.var_out, io.var, (size_t)B * d * sizeof(double));
        }
    }
    if (timers_env)
        fprintf(stderr, "[gprn] elbocalc_batch (N = %d, T = %d), %d evaluations, us: staging %.0f | set-up enqueued %.0f | %d sweeps: enqueue %.0f, "
                        "waiting for the device %.0f, verdicts %.0f | states back %.0f | total %.0f\n", N, w->T, B, us_stage, us_setup, n_sweeps,
                us_enqueue, us_wait, us_host, lap(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
    return GPRN_OK;
}

int mid_batch_elbocalc(gprn_ctx* c, int n_eval, const double* kparams, int n_kpar, const double* y_resid, const double* jitters,
                       const double* mu, const double* var, int max_iter, double* elbo, int* iters, int* conv, int* info,
                       double* mu_out, double* var_out)
{
    // (a one-tile problem with the small path switched off -- option "small_path" = 0, or gprn_keep_sigma on: there is no
    // batched form for it, as include/gprn_hip.h promises; inference.nELBO_batch then evaluates one by one -- ADVICE r5)
    if (c->T < 2) { c->err = "elbocalc_batch: one-tile problems run side by side on the small path only"; return GPRN_E_UNSUPPORTED; }
    if (c->comm || c->shm || c->world != 1) { c->err = "elbocalc_batch: one rank only"; return GPRN_E_UNSUPPORTED; }
    int kp_total = 0;
    for (int g = 0; g < c->G; ++g) {
        if (!c->kspec[g].set || c->kspec[g].uploaded) { c->err = "elbocalc_batch: every latent GP needs a device kernel program"; return GPRN_E_UNSUPPORTED; }
        kp_total += c->kspec[g].n_params;
    }
    if (kp_total != n_kpar) { c->err = "elbocalc_batch: kernel_params has the wrong length per evaluation"; return GPRN_E_ARG; }
    // (the budget is an estimate: when the device has less in one piece than it reports free, smaller chunks)
    int cap = 0, want = n_eval, rc_mem;
    while ((rc_mem = mid_ensure(c, want, &cap)) == GPRN_E_NOMEM && cap > 1) want = cap / 2;
    if (rc_mem) return rc_mem;
    c->err.clear();
    c->last_batch_chunk = std::min(cap, n_eval);
    MidBatch* m = (MidBatch*)c->mid_batch;
    gprn_ctx* w = m->w;
    // the worker follows the parent's switches
    w->use_flags = c->use_flags;
    w->wait_budget_ms = c->wait_budget_ms;
    w->overlap_opt = c->overlap_opt;
    w->acc_opt = c->acc_opt;
    w->fenced_finalize = c->fenced_finalize;
    w->pad_kb_opt = c->pad_kb_opt; w->pad_small_kb_opt = c->pad_small_kb_opt;
    w->prof.on = false;
    const size_t d = (size_t)(c->p + 1) * c->q * c->N, pn = (size_t)c->p * c->N;
    for (int e0 = 0; e0 < n_eval; e0 += cap) {
        const MidIo io{std::min(cap, n_eval - e0), kparams + (size_t)e0 * n_kpar, n_kpar, y_resid + (size_t)e0 * pn,
                       jitters + (size_t)e0 * c->p, mu + (size_t)e0 * d, var + (size_t)e0 * d, max_iter, elbo + e0, iters + e0,
                       conv + e0, info + e0, mu_out ? mu_out + (size_t)e0 * d : nullptr, var_out ? var_out + (size_t)e0 * d : nullptr};
        int rc = mid_chunk(c, m, io);
        if (rc == GPRN_E_WAIT_TIMEOUT) {
            // an in-kernel dependency wait gave up (a serialising tool, a starved device): both contexts go to the event
            // schedule and the chunk runs again from the caller's inputs (with_event_fallback's rule, api_internal.h)
            hipStreamSynchronize(w->stream); hipStreamSynchronize(w->stream2); hipStreamSynchronize(w->stream3);
            if (w->stream4) hipStreamSynchronize(w->stream4);
            w->use_flags = 0; c->use_flags = 0;
            c->fallbacks += 1;
            fprintf(stderr, "[gprn] elbocalc_batch: a device-side dependency wait timed out after %d ms%s; re-running the chunk with "
                            "HIP events (device-side waits are now off for this context)\n", w->wait_budget_ms, w->last_timeout.c_str());
            rc = mid_chunk(c, m, io);
            if (rc == GPRN_E_WAIT_TIMEOUT) { c->err = "elbocalc_batch: dependency wait timed out on the event schedule too"; rc = GPRN_E_HIP; }
        }
        if (rc) { if (rc < 0 && c->err.empty()) c->err = w->err; else if (rc < 0 && !w->err.empty() && c->err != w->err) c->err = w->err; return rc; }
    }
    return GPRN_OK;
}
