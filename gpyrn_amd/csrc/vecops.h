// Launchers of vecops.hip (all enqueue on ctx->stream against ctx->d_ptrs).
#pragma once
#include "gprn_internal.h"

int vec_prep(gprn_ctx* c, bool weights, const int* d_slot_gp, int nslots);
// part 0: all of B; 1 / 2: what the first outer panel (`outer` tiles) touches before its trailing update / the rest
int vec_build_B(gprn_ctx* c, int nslots, hipStream_t stream = nullptr, int part = 0, int outer = 0);
int vec_logdet(gprn_ctx* c, int buf, const int* d_slot_gp, int nslots, double* out);
// rows [row0, row0 + nrows) of the product (nrows < 0: to the last row)
int vec_lower_matvec(gprn_ctx* c, int buf, const double* vin, size_t vstride, int vin_by_gp,
                     const int* d_slot_gp, int nslots, double* out, hipStream_t stream = nullptr,
                     int row0 = 0, int nrows = -1);
int vec_colops(gprn_ctx* c, int nslots);                 // partial sums of every tile row + the reduction
int vec_colops_partial(gprn_ctx* c, int nslots, hipStream_t stream, int ch0, int nch);   // tile rows [ch0, ch0 + nch)
int vec_colops_reduce(gprn_ctx* c, int nslots);
int vec_reduce_finalize(gprn_ctx* c, const int* d_slot_gp, int nslots, bool with_logdet);   // vec_colops_reduce + vec_finalize in one launch
int vec_finalize(gprn_ctx* c, const int* d_slot_gp, int nslots, bool with_logdet = false);   // with_logdet: log det B from BUF_B too
int vec_q1(gprn_ctx* c, const double* Kinv_j, const double* Binv_k, const double* s_k,
           double* scratch, double* out_scalar, hipStream_t stream);
int vec_dot_self(gprn_ctx* c, const int* d_slot_gp, int nslots, const double* a, double* out, hipStream_t stream = nullptr);
int vec_elbo(gprn_ctx* c, double* out4, const double* scal, double* part, hipStream_t stream = nullptr);
#define GPRN_ELBO_PART_DOUBLES (3 * 32)
// several evaluations side by side (midn.hip): the ELBO assembly for the evaluations listed in d_evals, and the Q1 traces of
// their node slots (node-major: slot = k * n_eval + a; the traces go to c->d_q1 + evaluation * c->ev.scal)
int vec_elbo_evals(gprn_ctx* c, const int* d_evals, int n, double* out4, const double* scal, double* part, hipStream_t stream = nullptr);
int vec_q1_evals(gprn_ctx* c, const int* d_slot_eval, const double* Kinv_slab, int n_eval, double* scratch, hipStream_t stream = nullptr);
int vec_sigma(gprn_ctx* c, const double* Binv, const double* s, double* out);
int vec_pred_rows(gprn_ctx* c, int nslots, int ns, int ns_pad, const double* sol, const double* kss,
                  double* mean, double* var);
int vec_axpy_matrix(gprn_ctx* c, const double* src, double* dst, int N);   // dst += src on the N x N block (pitch ld)
int vec_symmetrize(gprn_ctx* c, double* M);                                 // upper := lower^T on the ld x ld matrix
// out4[l] = < 1/2 (P - Kinv + a a^T), dK/dtheta_l >, a = Kinv m, for a single SE / Periodic / QP kernel (kid, par[4])
int vec_grad_contract(gprn_ctx* c, int kid, const double* par, const double* Kinv, const double* P, const double* m,
                      double* a_scratch, double* part_scratch, double* out4);
// out = M v on the N x N block of an ld-pitched matrix (one wave per row)
int vec_symv(gprn_ctx* c, const double* M, const double* v, double* out);
