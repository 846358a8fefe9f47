// C ABI of libgprn_hip.so (include/gprn_hip.h): the set-up (meanfield.py:618-624), the sweep (ELBOaux, :651-710) and the
// ELBOcalc loop (:626-649), one evaluation or many side by side.
#include "api_internal.h"

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

