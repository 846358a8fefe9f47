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
//   diag   : L_kk = chol(B_kk), X_kk = L_kk^-1            one workgroup, LDS
//   panel  : L_ik = B_ik X_kk^T (i>k);  X_kc = X_kk R_kc (c<k)   tile GEMMs, K=128
//   update : B_ij -= L_ik L_jk^T (i>=j>k);  R_ic -= L_ik X_kc (i>k, c<=k)
// where R (the running right-hand side of L X = I) lives in X's own tiles and
// tile (i,c) is first written, not accumulated, at step k == c.  Per step the
// update touches (T-1-k)(T-k)/2 + (T-1-k)(k+1) tiles -- roughly constant until
// the tail, unlike POTRF alone.  Flops: N^3/3 + N^3/3.
#include "gprn_internal.h"

#include <math.h>

#define DPITCH 129

// ------------------------------------------------------------------ diag
// Unblocked potf2 + inverse of one 128x128 diagonal tile, in LDS.  The strict
// upper triangle of the LDS image holds the transposed running right-hand
// side: S[c][i] = R[i][c] (c < i).  Step k: v = column k of S scaled by
// 1/l_kk (rows i>k: L_ik; rows c<k: X_kc); then S[a][b] -= v'[a] v[b] for
// b > k, a in [0,k] U [b,127], with v'[k] = 1/l_kk.
__global__ __launch_bounds__(256)
void k_diag_block(double* const* __restrict__ ptrs, int ld, int kblk, int* __restrict__ info)
{
    extern __shared__ double S[];           // 128 x DPITCH, then dg[128]
    double* dg = S + 128 * DPITCH;
    const int slot = blockIdx.x;
    const size_t off = ((size_t)kblk * GPRN_TILE) * ld + (size_t)kblk * GPRN_TILE;
    double* Bt = ptrs[(size_t)slot * GPRN_NBUF + BUF_B] + off;
    double* Xt = ptrs[(size_t)slot * GPRN_NBUF + BUF_X] + off;
    const int tid = threadIdx.x;

    for (int e = tid; e < 128 * 128; e += 256) {
        const int r = e >> 7, c = e & 127;
        S[r * DPITCH + c] = (c <= r) ? Bt[(size_t)r * ld + c] : 0.0;
    }
    __syncthreads();

    const int b_lane = tid & 127, half = tid >> 7;
    for (int k = 0; k < 128; ++k) {
        __syncthreads();                      // step k-1's rank-1 update is complete
        const double piv = S[k * DPITCH + k];
        if (tid == 0 && !(piv > 0.0)) {
            if (info[slot] == 0) info[slot] = kblk * GPRN_TILE + k + 1;
        }
        const double lkk = sqrt(piv);         // NaN from here on for a non-PD input,
        const double inv = 1.0 / lkk;         // like jnp.linalg.cholesky
        if (tid < 128) {                      // nobody writes S[k][k] in this step
            if (tid != k) S[tid * DPITCH + k] *= inv;
            else dg[k] = lkk;
        }
        __syncthreads();
        const int b = k + 1 + b_lane;
        if (b < 128) {
            const double vb = S[b * DPITCH + k];
            if (half == 0) {
                for (int a = 0; a < k; ++a)
                    S[a * DPITCH + b] -= S[a * DPITCH + k] * vb;
                S[k * DPITCH + b] -= inv * vb;
            } else {
                for (int a = b; a < 128; ++a)
                    S[a * DPITCH + b] -= S[a * DPITCH + k] * vb;
            }
        }
    }
    __syncthreads();

    for (int e = tid; e < 128 * 128; e += 256) {
        const int r = e >> 7, c = e & 127;
        if (c < r) {
            Bt[(size_t)r * ld + c] = S[r * DPITCH + c];
            Xt[(size_t)r * ld + c] = S[c * DPITCH + r];
        } else if (c == r) {
            Bt[(size_t)r * ld + c] = dg[r];
            Xt[(size_t)r * ld + c] = 1.0 / dg[r];
        } else {
            Xt[(size_t)r * ld + c] = 0.0;
        }
    }
}

int launch_diag(gprn_ctx* c, double** d_ptrs, int nbatch, int ld, int kblk, int* d_info)
{
    static bool attr_set = false;
    const size_t shmem = (128 * DPITCH + 128) * sizeof(double);
    if (!attr_set) {
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_diag_block),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    prof_begin(c, GPRN_T_DIAG);
    hipLaunchKernelGGL(k_diag_block, dim3(nbatch), dim3(256), shmem, c->stream,
                       (double* const*)d_ptrs, ld, kblk, d_info);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ------------------------------------------------------------ task lists
static inline int64_t toff(int ti, int tj, int ld) {
    return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE;
}

int ensure_tasks(gprn_ctx* c)
{
    const int T = c->T, ld = c->ld;
    if (c->tasks_T == T && c->d_tasks) return GPRN_OK;
    std::vector<TileTask>& v = c->h_tasks;
    v.clear();
    c->steps.assign(T, gprn_ctx::StepRange{0, 0, 0, 0});
    for (int k = 0; k < T; ++k) {
        gprn_ctx::StepRange& s = c->steps[k];
        s.panel0 = v.size();
        for (int i = k + 1; i < T; ++i) {          // L_ik = B_ik X_kk^T   (in place)
            TileTask t{toff(i, k, ld), toff(i, k, ld), toff(k, k, ld), GPRN_TILE,
                       BUF_B, BUF_B, BUF_X, tile_modes(CM_SET, 0, 0)};
            v.push_back(t);
        }
        for (int cc = 0; cc < k; ++cc) {           // X_kc = X_kk R_kc     (in place)
            TileTask t{toff(k, cc, ld), toff(k, k, ld), toff(k, cc, ld), GPRN_TILE,
                       BUF_X, BUF_X, BUF_X, tile_modes(CM_SET, 0, 1)};
            v.push_back(t);
        }
        s.npanel = v.size() - s.panel0;
        s.upd0 = v.size();
        for (int i = k + 1; i < T; ++i) {
            for (int j = k + 1; j <= i; ++j) {     // B_ij -= L_ik L_jk^T
                TileTask t{toff(i, j, ld), toff(i, k, ld), toff(j, k, ld), GPRN_TILE,
                           BUF_B, BUF_B, BUF_B, tile_modes(CM_SUB, 0, 0)};
                v.push_back(t);
            }
            for (int cc = 0; cc <= k; ++cc) {      // R_ic -= L_ik X_kc ; first touch at c == k
                TileTask t{toff(i, cc, ld), toff(i, k, ld), toff(k, cc, ld), GPRN_TILE,
                           BUF_X, BUF_B, BUF_X,
                           tile_modes(cc == k ? CM_SETNEG : CM_SUB, 0, 1)};
                v.push_back(t);
            }
        }
        s.nupd = v.size() - s.upd0;
    }
    // lower(X^T X) -> BUF_B: tile (a,b), a >= b, sums over rows a*128 .. ld of X
    c->lauum0 = v.size();
    for (int a = 0; a < T; ++a)
        for (int b = 0; b <= a; ++b) {
            TileTask t{toff(a, b, ld), toff(a, a, ld), toff(a, b, ld), ld - a * GPRN_TILE,
                       BUF_B, BUF_X, BUF_X, tile_modes(CM_SET, 1, 1)};
            v.push_back(t);
        }
    c->nlauum = v.size() - c->lauum0;

    if (v.size() > c->tasks_cap) {
        if (c->d_tasks) hipFree(c->d_tasks);
        c->d_tasks = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_tasks, v.size() * sizeof(TileTask)));
        c->tasks_cap = v.size();
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_tasks, v.data(), v.size() * sizeof(TileTask),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tasks_T = T;
    return GPRN_OK;
}

int factor_invert(gprn_ctx* c, int nbatch)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    for (int k = 0; k < c->T; ++k) {
        const gprn_ctx::StepRange& s = c->steps[k];
        if ((rc = launch_diag(c, c->d_ptrs, nbatch, c->ld, k, c->d_info_cur))) return rc;
        if ((rc = launch_tiles(c, c->d_tasks + s.panel0, s.npanel, c->d_ptrs, nbatch, c->ld,
                               GPRN_T_PANEL))) return rc;
        if ((rc = launch_tiles(c, c->d_tasks + s.upd0, s.nupd, c->d_ptrs, nbatch, c->ld,
                               GPRN_T_UPDATE))) return rc;
    }
    return GPRN_OK;
}

// BUF_B of every slot = lower(X^T X), X in BUF_X (L in BUF_B is overwritten)
int lauum_lower(gprn_ctx* c, int nbatch)
{
    int rc = ensure_tasks(c);
    if (rc) return rc;
    return launch_tiles(c, c->d_tasks + c->lauum0, c->nlauum, c->d_ptrs, nbatch, c->ld,
                        GPRN_T_LAUUM);
}
