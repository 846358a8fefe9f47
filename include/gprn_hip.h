/* gprn_hip.h -- C ABI of libgprn_hip.so, the MI355X (gfx950) backend of the
 * gpyrn mean-field ELBO hot path.
 *
 * The reference (iastro-pt/gpyrn) is pure Python and has no FFI; the boundary
 * this library sits behind is `gpyrn/meanfield.py`'s `inference.ELBOcalc` /
 * `ELBOaux` (meanfield.py:561-710) and the kernel-matrix assembly
 * `inference._KMatrix` (meanfield.py:413-434) over `gpyrn/covfunc.py`.  Each
 * entry point below names the reference code it replaces.  The only caller is
 * gpyrn_amd/_hip.py (ctypes); INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every function returns int: 0 ok; >0 a LAPACK-style `info` (order of the
 *    first non-positive pivot, see gprn_last_info_gp); <0 GPRN_E_* below, with
 *    text in gprn_last_error().
 *  - all arrays are C-contiguous IEEE fp64 host buffers owned by the caller and
 *    only touched during the call; device memory is owned by the context.
 *  - one context = one GPU = one host thread at a time.  Create contexts after
 *    fork()/in spawned workers (HIP state does not survive fork).
 *  - latent GP index `gp`: 0..q-1 are the nodes, q + (j*p + i) is the weight
 *    of node j / output i  (the reference's flat Kw order, meanfield.py:620,749).
 */
#ifndef GPRN_HIP_H
#define GPRN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gprn_ctx gprn_ctx;

enum {
    GPRN_OK = 0,
    GPRN_E_ARG = -1,      /* bad argument / call order */
    GPRN_E_HIP = -2,      /* HIP runtime error */
    GPRN_E_NODEV = -3,    /* no usable GPU */
    GPRN_E_COMM = -4,     /* RCCL error */
    GPRN_E_NOMEM = -5,
    GPRN_E_UNSUPPORTED = -6   /* the request is well-formed but this object has no device form for it (see the entry point) */
};

/* ---- kernel ids of the fused covariance fill (covfunc.py line numbers) ---- */
enum {
    GPRN_K_CONSTANT = 0,            /* :123-125 */
    GPRN_K_WHITENOISE = 1,          /* :144-148, square-matrix branch */
    GPRN_K_SE = 2,                  /* :169-170 */
    GPRN_K_PERIODIC = 3,            /* :211-213 */
    GPRN_K_QP = 4,                  /* :251-255 */
    GPRN_K_RQ = 5,                  /* :286-288 */
    GPRN_K_RQP = 6,                 /* :310-313 */
    GPRN_K_COSINE = 7,              /* :330-331 */
    GPRN_K_EXPONENTIAL = 8,         /* :351-352 */
    GPRN_K_MATERN32 = 9,            /* :370-373 */
    GPRN_K_MATERN52 = 10,           /* :391-396 */
    GPRN_K_GAMMAEXP = 11,           /* :431-432 */
    GPRN_K_PIECEWISE = 12,          /* :469-473 */
    GPRN_K_PACIOREK = 13,           /* :493-496 */
    GPRN_K_NEWPERIODIC = 14,        /* :517-519 */
    GPRN_K_QUASINEWPERIODIC = 15,   /* :543-546 */
    GPRN_K_COSPERIODIC = 16,        /* :664-665 */
    GPRN_K_QUASICOSPERIODIC = 17,   /* :687-689 */
    GPRN_K_POLYNOMIAL = 18,         /* :454-455, two-argument */
    GPRN_K_HARMONICPERIODIC = 19,   /* :598-607, two-argument */
    GPRN_K_QUASIHARMONICPERIODIC = 20, /* :631-642, two-argument */
    GPRN_K_DSE = 21,                /* :182-185 */
    GPRN_K_DPERIODIC = 22,          /* :215-221 */
    GPRN_K_DQP = 23,                /* :257-266 */
    GPRN_K_COUNT = 24
};
/* postfix opcodes of a kernel expression (covfunc.py:65-77 Sum/Multiplication) */
enum { GPRN_OP_PUSH = 0, GPRN_OP_ADD = 1, GPRN_OP_MUL = 2 };
#define GPRN_MAX_OPS 32
#define GPRN_MAX_KPARAMS 64

/* ---- context ---- */
int gprn_device_count(void);
/* Replaces nothing in the reference (it has no device).  One context per model; contexts on one device share the
 * library's streams (one set per device and process) and every call below is synchronous and takes the device's
 * lock, so contexts may be used from several host threads -- their calls run one after the other. */
int gprn_create(gprn_ctx** out, int device_id);
void gprn_destroy(gprn_ctx* ctx);
const char* gprn_last_error(const gprn_ctx* ctx);
/* which latent GP the last positive `info` belongs to (-1 if none) */
int gprn_last_info_gp(const gprn_ctx* ctx);

/* ---- problem: inference.__init__ data layout, meanfield.py:106-134 ----
 * y, yerr are (p, N) row-major: the reference's self.y / self.yerr. */
int gprn_set_data(gprn_ctx* ctx, int N, int p, int q,
                  const double* time, const double* y, const double* yerr);

/* ---- multi-GPU sharding (new): one context per rank/GPU.
 * comm_init before set_data; set_owners after set_data and before set_kernel:
 * latent GP g is factored and updated by rank owner[g] (q + q*p entries).
 * Transport: RCCL (ncclUniqueId in id128).  With GPRN_COMM_TRANSPORT=shm in the
 * environment comm_unique_id returns the name of a host shared-memory segment instead
 * and the same three collectives (row broadcast, scalar all-reduce, barrier) cross it by
 * host copies: a rehearsal transport so that several ranks can share ONE GPU in tests
 * (RCCL refuses that); not a production path. */
int gprn_comm_unique_id(char* id128);
int gprn_comm_init(gprn_ctx* ctx, int world, int rank, const char* id128);
int gprn_set_owners(gprn_ctx* ctx, const int* owner);
int gprn_comm_barrier_max(gprn_ctx* ctx, double* value); /* all-reduce(max) + sync */
/* sum of a host vector over the ranks, result on every rank (pool of independent ELBO
 * evaluations across GPUs: meanfield.py:1222-1260 evaluates its walkers one by one) */
int gprn_comm_allreduce_sum(gprn_ctx* ctx, double* buf, int n);

/* ---- per-ELBOcalc setup: meanfield.py:618-624 ----
 * set_kernel: latent GP `gp` gets K = expr(t_i, t_j) (+ 1e-6 I when add_nugget,
 * meanfield.py:432-433); ops = n_ops triples (opcode, kernel id, param offset).
 * upload_K: K evaluated by the caller (user-defined covFunction subclasses). */
int gprn_set_kernel(gprn_ctx* ctx, int gp, const int32_t* ops, int n_ops,
                    const double* params, int n_params, int add_nugget);
int gprn_upload_K(gprn_ctx* ctx, int gp, const double* K);
int gprn_set_y_resid(gprn_ctx* ctx, const double* y_minus_mean);   /* (p,N), :623-624 */
int gprn_set_jitters(gprn_ctx* ctx, const double* jitters);        /* (p), :618 */
/* covariance fill + chol(K) (+ K^-1 pieces the sweep needs): replaces
 * _KMatrix (:413-434) and _cholNugget (:71-89) of the setup block :619-622. */
int gprn_factor_priors(gprn_ctx* ctx);

/* ---- variational state: mu/var in the reference's flat layout (d = N q (p+1)),
 * meanfield.py:473-489 ---- */
int gprn_set_muvar(gprn_ctx* ctx, const double* mu, const double* var);
int gprn_get_muvar(gprn_ctx* ctx, double* mu, double* var);

/* ---- the hot loop: n_sweeps x ELBOaux (meanfield.py:651-710 = _updateSigMu
 * :713-893 + _entropy :1069-1093 + _expectedLogPrior :992-1067 +
 * _expectedLogLike :895-990).  elbo_out[n_sweeps]; parts_out[3*n_sweeps] =
 * (LogL, LogP, Ent) per sweep, may be NULL.  commit=0 evaluates the sweep but
 * leaves mu/var untouched (ELBOcalc's discarded first call, :627). */
int gprn_sweep(gprn_ctx* ctx, int n_sweeps, int commit,
               double* elbo_out, double* parts_out);

/* ---- prediction: conditional mean and variance of every latent GP at `ns` new
 * times from the current variational state (gprn_set_muvar or the last sweep): replaces
 * _gp.GP.prediction (_gp.py:107-138) under inference._Prediction (meanfield.py:1289-1381).
 * mean_out, var_out: (q + q*p, ns) row-major, row = latent GP index (all of them on every rank: on a sharded context
 * the owners' rows are broadcast).  Kernels set with gprn_set_kernel are filled on the device; for matrices that came
 * through gprn_upload_K see gprn_predict_upload. */
int gprn_predict(gprn_ctx* ctx, int ns, const double* tstar, double* mean_out, double* var_out);
/* The same for a latent GP whose covariance is a user-defined covFunction subclass (its K came through gprn_upload_K):
 * the caller evaluates, for the NEXT gprn_predict call with this `ns`, what the reference evaluates in Python --
 * K_tiny = kernel(t_i - t_j) + 1.25e-12 I (N, N; _gp.GP._kernel_matrix, _gp.py:40-50 = inference._tinyNuggetKMatrix,
 * meanfield.py:436-452), Kstar = kernel(tstar_i - t_j) (ns, N; _gp.py:52-63 = _predictKMatrix, meanfield.py:455-471)
 * and kss[i] = the diagonal of _kernel_matrix(kernel, tstar) (ns).  The factorisation and the solves stay on the GPU.
 * On a sharded context only the owner of `gp` keeps the matrices, and gprn_predict returns every latent GP's rows on
 * every rank (the owners' results travel as one grouped broadcast). */
int gprn_predict_upload(gprn_ctx* ctx, int gp, int ns, const double* K_tiny, const double* Kstar, const double* kss);

/* ---- kernel matrices and prior draws outside the ELBO loop ----
 * eval_kernel: K = expr(t_i, t_j) + nugget I at the data times through the fused fill kernel: replaces
 * inference._KMatrix (meanfield.py:413-434; nugget 1e-6) and _tinyNuggetKMatrix (:436-452; 1.25e-12) when
 * they are called on their own.  K_out (N, N).
 * sample_prior: out[s] = L z[s], K + nugget I = L L^T by the blocked factorisation: replaces
 * inference._sample_from_gp / sample (:517-539), which draw from scipy's multivariate_normal.  z, out:
 * (n_samples, N); z = standard normals of the caller's generator.  Returns info > 0 when K + nugget I is
 * not positive definite in fp64. */
int gprn_eval_kernel(gprn_ctx* ctx, const int32_t* ops, int n_ops, const double* params, int n_params,
                     double nugget, double* K_out);
int gprn_sample_prior(gprn_ctx* ctx, const int32_t* ops, int n_ops, const double* params, int n_params,
                      double nugget, int n_samples, const double* z, double* out);

/* ---- analytic gradient of the ELBO in the kernel hyper-parameters (not in the reference,
 * whose optimiser is derivative-free, meanfield.py:1149-1150; the derivative hooks it carries are
 * covfunc.py:172-185, 215-221, 257-266).  At fixed variational state only the expected log prior
 * (meanfield.py:992-1067) depends on K_gp:  d/dtheta = 1/2 < K^-1 S K^-1 + a a^T - K^-1, dK/dtheta >, with S the
 * covariance the reference pairs with K_gp (node j: Sigma_f0 + ... + Sigma_fj; weight: its Sigma_w) and
 * a = K^-1 m.  This entry does the O(N^3) part on the device and returns K^-1 and P = K^-1 S K^-1, both
 * (N, N) symmetric; needs gprn_factor_priors and a committed sweep with gprn_keep_sigma(1).  Unsharded
 * contexts only. */
int gprn_grad_matrices(gprn_ctx* ctx, int gp, double* Kinv_out, double* P_out);
/* the contraction as well on the device: grad_out[l] = < 1/2 (K^-1 S K^-1 + a a^T - K^-1), dK/dtheta_l >, l < n_params,
 * a = K^-1 m (m: N values, the mean the reference pairs with that kernel; the 1/q of meanfield.py:709 is left to the
 * caller).  dK/dtheta in closed form for a single SquaredExponential, Periodic or QuasiPeriodic (the formulas of
 * covFunction._dk_dpars), by central differences of the kernel program (relative step 1e-6) for every other kernel
 * gprn_set_kernel accepted.  GPRN_E_UNSUPPORTED for a latent GP whose matrix was uploaded (gprn_upload_K). */
int gprn_grad_kernel(gprn_ctx* ctx, int gp, const double* m, double* grad_out);

/* ---- the terms of the ELBO on their own: what the reference's private step methods return (meanfield.py:895-990 and
 * 992-1067; ELBOaux :651-710 calls them in turn, and scripts written against the reference may too).
 * expected_loglike: inference._expectedLogLike of the state last set (gprn_set_muvar: var = the diagonals of Sigma_f, Sigma_w)
 * under the jitters last set (gprn_set_jitters); the raw data enter as in the reference (quirk Q3).
 * prior_terms: for latent GP `gp` and a covariance S (N, N) / mean m (N) of the caller's choosing -- the reference pairs node j
 * with the cumulative Sigma_f0 + ... + Sigma_fj and weight (j, i) with the raw-reshape row of mu_w (quirks Q1, Q2) -- from the
 * factor of K_gp that gprn_factor_priors left on the device: out3 = { log det K_gp, m^T K_gp^-1 m, tr(K_gp^-1 S) }
 * (:1029-1041, 1050-1062).  Unsharded contexts. */
int gprn_expected_loglike(gprn_ctx* ctx, double* logl_out);
int gprn_prior_terms(gprn_ctx* ctx, int gp, const double* S, const double* m, double* out3);

/* ---- read-back for tests and the ELBOaux compatibility shim ---- */
enum {
    GPRN_M_K = 0,        /* prior covariance K_gp (N,N) */
    GPRN_M_KLINV = 1,    /* chol(K_gp)^-1, lower */
    GPRN_M_SIGMA = 2,    /* variational covariance of the last sweep (N,N) */
    GPRN_M_BX = 3,       /* X = chol(B)^-1 (lower) of the last half-sweep that factored this latent GP, B = I + D^1/2 K D^1/2:
                            what the posterior variances are column sums of; diagnostics */
    GPRN_M_BL = 4        /* ... and chol(B) itself (lower); for a node k < q - 1 with q > 1 it has been overwritten by
                            lower(B^-1) (quirk Q1) */
};
int gprn_keep_sigma(gprn_ctx* ctx, int on);   /* form Sigma explicitly during sweeps (ELBOaux shim) */
int gprn_get_matrix(gprn_ctx* ctx, int which, int gp, double* out);
int gprn_get_logdet_K(gprn_ctx* ctx, double* out /* q+q*p */);
/* per-GP scalars of the last sweep: log det B [G], tr(B^-1) [G], m^T K^-1 m [G], <K_j^-1, Sigma_k> [q*q]:
 * what the entropy (meanfield.py:1069-1093) and the prior term (:992-1067) are assembled from; G = q + q*p */
int gprn_get_scalars(gprn_ctx* ctx, double* out /* 3 G + q*q */);

/* ---- timing hooks used by bench.py (HIP events on the library's stream) ----
 * milliseconds spent in, and launches of, each kernel family since the last
 * reset; only collected while profiling is enabled (adds event records). */
enum { GPRN_T_FILL = 0, GPRN_T_BUILD_B = 1, GPRN_T_DIAG = 2, GPRN_T_PANEL = 3,
       GPRN_T_UPDATE = 4, GPRN_T_LAUUM = 5, GPRN_T_VEC = 6,
       GPRN_T_UPDATE_AHEAD = 7,   /* the part of a trailing update the next panel's update writes again (own launch) */
       GPRN_T_COUNT = 8 };
int gprn_profile_enable(gprn_ctx* ctx, int family_mask);   /* bit f = time family GPRN_T_f; 0 = off */
int gprn_profile_read(gprn_ctx* ctx, double* ms /*GPRN_T_COUNT*/,
                      int64_t* launches /*GPRN_T_COUNT*/, int reset);

/* inference.ELBOcalc (meanfield.py:561-649) from its set-up block on, in one call.
 *   do_setup != 0: first the set-up of gprn_factor_priors (:618-622) with the kernels last given by gprn_set_kernel /
 *       gprn_upload_K; 0: the factors of the last set-up are kept (unchanged hyper-parameters).
 *   y_resid (p x N, y minus the mean functions, :624), jitters (p), mu / var (the state the loop starts from, d each):
 *       as gprn_set_y_resid / gprn_set_jitters / gprn_set_muvar; NULL keeps what was set before.
 * Then the loop of :626-649: one sweep whose update is discarded and whose ELBO is kept as elboArray[0] (:627-628),
 * committed sweeps until `iterNumber > 3 and |std(last3) / mean(last3)| < 1e-3 and != 0` (:640-643, np.std = population
 * std) or max_iter trips.  history[0 .. *n_history) receives elboArray (if it is longer than cap, its last cap values),
 * *iterations the trip count, *converged whether the stop rule fired, mu_out / var_out (d each, or both NULL) the state
 * the loop ended in (it also stays on the device: gprn_get_muvar).  Returns as gprn_sweep; a pivot failure of the set-up
 * is reported the same way.
 * Problems of one tile (N <= 128) do all of this with ONE host synchronisation per eight sweeps (csrc/smalln.hip: inputs
 * through pinned staging and asynchronous copies, the loop and its stop rule on the device, sweeps enqueued ahead of
 * the verdict); larger ones run the entry points above in turn. */
int gprn_elbocalc(gprn_ctx* ctx, int do_setup, const double* y_resid, const double* jitters, const double* mu,
                  const double* var, int max_iter, double* history, int cap, int* n_history, int* iterations,
                  int* converged, double* mu_out, double* var_out);

/* (On a sharded context every local finding of gprn_elbocalc -- arguments, call order, a failing setter -- is agreed between
 * the ranks before its first collective: either every rank goes on or every rank returns.) */

/* n_eval INDEPENDENT evaluations of the same problem at n_eval parameter vectors -- what scipy's simplex or emcee's
 * walkers ask inference.nELBO for one after the other (meanfield.py:1095-1111, 1222-1260) -- side by side on the device:
 * every launch covers all of them, each evaluation with its own covariance matrices, factors, state, loop and stop rule.
 *   kernel_params [n_eval][n_kernel_params]: the parameters of every latent GP's kernel, concatenated in latent-GP order,
 *       for the kernel PROGRAMS last given by gprn_set_kernel (same expression trees, other values);
 *   y_resid [n_eval][p N], jitters [n_eval][p], mu / var [n_eval][d]: per evaluation, as for gprn_elbocalc.
 * Out per evaluation: the last ELBO of its loop, its trip count, whether the stop rule fired, its info (> 0: a pivot
 * failed, the ELBO is NaN), and (or both NULL) the state it ended in, [n_eval][d].
 * One rank, device kernels only (every latent GP's kernel given by gprn_set_kernel, even in t_i - t_j): GPRN_E_UNSUPPORTED
 * otherwise (the caller evaluates one by one).  One-tile problems (N <= 128) run a half-sweep of ALL evaluations as one
 * launch (csrc/smalln.hip); larger ones go through the launch schedule of the large problems with its batch dimension =
 * evaluations x latent GPs of the phase, an evaluation that has stopped leaving the next sweep's launches (csrc/midn.hip).
 * Lists longer than the memory budget (option "batch_mem_mb") run chunk by chunk.  The context's own state and factors
 * are not touched.  An evaluation whose factorisation fails returns info > 0 and a NaN ELBO at once (the reference's loop
 * would carry the NaN to max_iter: iterations reports max_iter). */
int gprn_elbocalc_batch(gprn_ctx* ctx, int n_eval, const double* kernel_params, int n_kernel_params,
                        const double* y_resid, const double* jitters, const double* mu, const double* var,
                        int max_iter, double* elbo, int* iterations, int* converged, int* info,
                        double* mu_out, double* var_out);

/* ---- per-context switches (tests, experiments; nothing in the reference corresponds) ----
 * name: "flags" (1: the factorisation's cross-stream dependencies travel through device-side flags and
 * in-kernel waits, 0: HIP events -- chosen automatically per context, and latched to 0 after an in-kernel
 * wait timed out, in which case the call is re-run on events); "wait_budget_ms" (wall-clock budget of one
 * in-kernel wait); "withhold_inner" (test hook: the n-th in-panel completion flag of every following call is
 * never raised); "fallbacks" (read-only count of re-run calls); "bulk_pad_kb" / "small_pad_kb" (KiB of unused
 * dynamic LDS the bulk tile launches -- batches above / up to two matrices -- ask for, to keep CUs open for the
 * latency chain; -2 returns to the default; a pad that does not fit a workgroup's LDS makes the factorising calls
 * return GPRN_E_ARG instead of aborting the queue); "overlap" (bit mask of what runs beside the factorisations
 * instead of before / behind them: 1 B formed inside the first panel's update, 2 row reductions over X panel by
 * panel, 4 node term beside the weight phase, 8 log det B in the finalising kernel, 16 a sweep's end beside the
 * next sweep's node phase; results are bit-identical for every value); "small_path" (1, the default: problems of
 * one tile -- N <= 128 -- run each half-sweep as ONE launch, one workgroup per latent GP, csrc/smalln.hip; 2: problems
 * of two tiles too; 0: the launch schedule at every size; same results to rounding); "batch_mem_mb" (device memory, MiB, that
 * one chunk of gprn_elbocalc_batch's evaluations may take: longer lists run chunk by chunk; default: half of what is free, 48 GiB
 * at most; when the device cannot give that much in one piece the chunk is halved until it can); "batch_chunk" (read-only:
 * evaluations per chunk in the last gprn_elbocalc_batch call); "comm_budget_s" (sharded contexts: seconds an entry point may stay inside its collective section -- a rank that
 * died leaves the others there -- before the library's watchdog names the entry point, the collective and the rank on
 * stderr and ends the process with status 86; default 600, or GPRN_COMM_BUDGET_S); "accurate_factor" (the panel steps of
 * a blocked factorisation as triangular SOLVES -- what LAPACK's potrf does -- instead of products with the explicit inverse
 * of the diagonal block: by default (-2 returns to it) in every factorisation of a PRIOR matrix, i.e. the set-up
 * (meanfield.py:71-89, 621-622), prediction and prior draws, where cond(K) ~ 1e8 under the reference's 1e-6 nugget and a
 * product costs eps cond(K) on m^T K^-1 m (meanfield.py:1032, 1050); 0: never; 1: in the sweeps of the launch path as well).
 * "fenced_finalize" (test hook: the finalising kernel of a phase hands its partial terms to its last workgroup with
 * release / acquire fences instead of the gfx942 / gfx950 shortcut documented in csrc/vecops.hip; same bits).
 * value == -1 only reads; *old (may be
 * NULL) receives the previous value. */
int gprn_set_option(gprn_ctx* ctx, const char* name, int value, int* old);

/* ---- diagnostic entry points: one kernel each, for tests/test_kernels_gpu.py ----
 * C (+)= A.B on host matrices through the MFMA tile kernel; modes as in
 * csrc/gemm_tile.hip (a_mode 0: A[m][k], 1: A[k][m]; b_mode 0: B[n][k], 1: B[k][n];
 * c_mode 0: C=AB, 1: C-=AB, 2: C=-AB; bits 4-5 of c_mode pick the workgroup shape 128x128, 64x64,
 * 64x128, 128x64).  M, N multiples of 128; K multiple of 16. */
int gprn_test_gemm(gprn_ctx* ctx, int M, int N, int K, int a_mode, int b_mode,
                   int c_mode, const double* A, const double* B, double* C);
/* time (ms, average of reps) of C -= A.B^T, M x N x K on random device data, through the tile contraction: one launch
 * with 64 x 64 (how 0) / 128 x 128 (how 1) workgroups */
int gprn_test_gemm_rate(gprn_ctx* ctx, int M, int N, int K, int how, int reps, double* ms);
/* time (ms per pass, average of reps) of the set-up's covariance fills -- every local latent GP's kernel into its K,
 * launch behind launch -- inside one pair of events: the rate the fill kernels themselves run at */
int gprn_test_fill_rate(gprn_ctx* ctx, int reps, double* ms);
/* in: SPD A (n x n, n multiple of 128); out: L (lower, upper zeroed) and L^-1 */
int gprn_test_factor_invert(gprn_ctx* ctx, int n, int batch, const double* A,
                            double* L, double* Linv);
/* back-to-back v_mfma_f64_16x16x4_f64 from registers on every CU: the measured fp64 MFMA
 * ceiling of this device in TFLOP/s (what roofline fractions can be judged against) */
int gprn_test_mfma_peak(gprn_ctx* ctx, int wg_per_cu, int iters, double* tflops);
/* out = lower(X^T X) for lower-triangular X */
int gprn_test_lauum(gprn_ctx* ctx, int n, const double* X, double* out);

#ifdef __cplusplus
}
#endif
#endif /* GPRN_HIP_H */
