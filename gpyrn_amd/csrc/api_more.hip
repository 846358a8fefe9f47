// C ABI of libgprn_hip.so (include/gprn_hip.h): prediction (meanfield.py:1289-1400), kernel matrices and prior draws (:413-434,
// 517-539), the gradient's pieces, the ELBO's terms on their own (:895-1093), diagnostics.
#include "api_internal.h"

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
        if ((rc = comm_group(c, true))) goto done;
        for (int g = 0; g < c->G && !rc; ++g) {
            rc = comm_broadcast(c, d_all + (size_t)g * ns, ns, c->owner[g]);
            if (!rc) rc = comm_broadcast(c, d_all + ((size_t)c->G + g) * ns, ns, c->owner[g]);
        }
        { const int rg = comm_group(c, false); if (!rc) rc = rg; }
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

int test_setup(gprn_ctx* c, int ld, int nbuf_needed, int batch);

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
int test_setup(gprn_ctx* c, int ld, int nbuf_needed, int batch)
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

