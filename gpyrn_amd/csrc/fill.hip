// Fused pairwise-difference + covariance fill:  K[m][n] = k(t_m, t_n) (+ nugget).
//
// Replaces inference._KMatrix (meanfield.py:413-434) over covFunction.__call__
// (covfunc.py, line numbers per kernel in include/gprn_hip.h): the reference
// materialises r = t[:,None]-t[None,:] and 5-8 more N x N temporaries per
// kernel; here each element is produced from two reads of the time vector
// (L2 resident) and written once -- 8 N^2 bytes of HBM traffic per matrix.
//
// A kernel expression (Sum / Multiplication trees of built-ins, covfunc.py:65-77)
// arrives as a postfix program evaluated per element on a tiny register stack.
// Formulas follow the reference's operation order so host and device agree to
// rounding (device libm vs NumPy: <= 2 ulp).
#include "gprn_internal.h"

#include <math.h>
#include <string.h>

struct FillProgram {
    int n_ops;
    int nugget;
    double nugget_val;        // 1e-6 (meanfield.py:433) for the priors, 1.25e-12 (_gp.py:47) for prediction
    int32_t ops[3 * GPRN_MAX_OPS];
    double par[GPRN_MAX_KPARAMS];
    double aux[3];            // one-kernel SE / Periodic / QP programs: the reciprocals the element formula multiplies by
};

#define PI_D 3.141592653589793
// GPRN_FILL_FAST=0 at build time: the device library's exp / sinpi in the SE, Periodic and QP kernels (rounds 1-2)
#ifndef GPRN_FILL_FAST
#define GPRN_FILL_FAST 1
#endif

// exp(x) for x <= 0 -- the exponent of every kernel below is one: Cody-Waite reduction by ln 2 in two words, the Taylor
// polynomial of degree 13 on |r| <= ln(2)/2 (truncation 4e-18), one ldexp (which also rounds into the denormals and
// flushes to zero below them).  No branches, no special cases: 2.6e-16 worst relative error against long double on the
// host over the benchmark's arguments, the same as the device library's exp, in 19 instead of ~30 instructions.
__device__ __forceinline__ double exp_neg(double x)
{
    const double x_in = x;
    x = fmax(x, -800.0);
    const double n = rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // (a NaN argument -- a NaN hyper-parameter -- stays NaN as in NumPy: fmax above would have turned it into exp(-800) = 0)
    return x_in != x_in ? x_in : ldexp(p, (int)n);
}

// a / b correctly rounded (but for a sliver of near-halfway cases) from rb = RN(1 / b): one residual, one correction -- two
// FMAs where the IEEE division sequence is ~10 quarter-rate instructions.  The reference divides (NumPy: x / ell**2), and on
// a prior matrix with cond(K) ~ 1e9 the last bits of K's entries are worth 1e-8 on m^T K^-1 m (profiles/r06_fill_rounding.txt).
__device__ __forceinline__ double div_rn(double a, double b, double rb)
{
    const double q = a * rb;
    return fma(fma(-q, b, a), rb, q);
}

// sin^2(x), x >= 0 in RADIANS -- the periodic kernels as the reference writes them, np.sin(np.pi * np.abs(r) / P)**2: the
// argument is the ROUNDED product / quotient, several hundred periods out, so its rounding error (|x| 1.1e-16 absolute) is part
// of the reference's value: sinpi_sq of the exact fraction |r| / P is closer to the mathematical kernel but differs from
// NumPy's by tens to hundreds of ulp of K (mean 34, max 589 at |x| ~ 110), which the Cholesky of an ill-conditioned prior
// amplifies.  Reduction by pi in three words (Cody-Waite: the first two have 33 bits, n pi_A and n pi_B are exact for
// n < 2^19 -- |x| < 1.6e6, i.e. 5e5 periods), then sin on [0, pi/4] by its Taylor polynomial of degree 19 (next term 8e-20)
// and, beyond pi/4, 1 - sin^2(pi/2 - g).
__device__ __forceinline__ double sin_sq_rad(double x)
{
    const double n = rint(x * 0.318309886183790671538);
    double g = fma(-n, 3.14159265346825122833e+00, x);
    g = fma(-n, 1.21542010126079319532e-10, g);
    g = fma(-n, 4.04453249742233291160e-21, g);
    g = fabs(g);
    const bool hi = g > 0.78539816339744830962;
    const double u = hi ? (1.57079632679489655800e+00 - g) + 6.12323399573676603587e-17 : g;
    const double z = u * u;
    double p = -1.0 / 121645100408832000.0;
    p = fma(p, z, 1.0 / 355687428096000.0);
    p = fma(p, z, -1.0 / 1307674368000.0);
    p = fma(p, z, 1.0 / 6227020800.0);
    p = fma(p, z, -1.0 / 39916800.0);
    p = fma(p, z, 1.0 / 362880.0);
    p = fma(p, z, -1.0 / 5040.0);
    p = fma(p, z, 1.0 / 120.0);
    p = fma(p, z, -1.0 / 6.0);
    const double sn = fma(p * z, u, u);
    const double s2 = sn * sn;
    return hi ? 1.0 - s2 : s2;
}

__device__ __forceinline__ void harmonic_terms(double Nh, double P, double t, double& s, double& u)
{
    // covfunc.py:599-605 with its precedence: sin(phase)/2*sin(half)
    const double phase = (Nh + 0.5) * 2 * PI_D * t / P;
    const double half = PI_D * t / P;
    s = sin(phase) / 2 * sin(half);
    u = 0.5 / tan(half) - cos(phase) / 2 * sin(half);
}

__device__ __forceinline__ double eval_kernel(int kid, const double* __restrict__ q,
                                              double ti, double tj, bool diag, const double* __restrict__ aux = nullptr)
{
    const double r = ti - tj;
    switch (kid) {
    case GPRN_K_CONSTANT: return q[0] * q[0];
    case GPRN_K_WHITENOISE: return diag ? q[0] * q[0] : 0.0;
    // SE, Periodic, QP (the kernels of the BASELINE configs): the per-element divisions by parameter expressions run as
    // multiplications by reciprocals -- which depend on the parameters only, are formed once on the host (aux, make_program)
    // and hoisted out of the element loop: an IEEE fp64 division is ~10 quarter-rate instructions, and QP's three were a
    // quarter of a thread's instructions -- plus ONE correction step each (div_rn), so that the quotients round as NumPy's
    // divisions do; the sine takes the reference's ROUNDED radian argument (sin_sq_rad).  Rounds 1-5 multiplied by the
    // reciprocal alone and took sin^2(pi frac(|r| / P)): closer to the mathematical kernel, tens to hundreds of ulp away
    // from the reference's K -- which a prior with cond(K) ~ 1e9 turns into 1e-8 on the ELBO (profiles/r06_fill_rounding.txt).
    case GPRN_K_SE: {                 // theta**2 * exp(-0.5 * r**2 / ell**2)
        const double l2 = q[1] * q[1];
        const double x = GPRN_FILL_FAST ? div_rn(-0.5 * (r * r), l2, aux ? aux[0] : 1.0 / l2) : -0.5 * (r * r) / l2;
        return q[0] * q[0] * (GPRN_FILL_FAST ? exp_neg(x) : exp(x));
    }
    // (sin(pi |r| / P) through the fraction of |r| / P, not through sin on an argument of several hundred periods, whose
    // reduction is a multi-word product)
    case GPRN_K_PERIODIC: {           // theta**2 * exp(-2 * sin(pi * |r| / P)**2 / ell**2)
        const double l2 = q[2] * q[2];
        double x;
        if (GPRN_FILL_FAST) {
            const double s2 = sin_sq_rad(div_rn(PI_D * fabs(r), q[1], aux ? aux[1] : 1.0 / q[1]));
            x = div_rn(-2 * s2, l2, aux ? aux[0] : 1.0 / l2);
        } else { const double sn = sin(PI_D * fabs(r) / q[1]); x = -2 * (sn * sn) / l2; }
        return q[0] * q[0] * (GPRN_FILL_FAST ? exp_neg(x) : exp(x));
    }
    case GPRN_K_QP: {                 // theta**2 * exp(-2 * sin(pi * |r| / P)**2 / ellp**2 - r**2 / (2 * elle**2))
        const double lp2 = q[3] * q[3], le2 = 2 * (q[1] * q[1]);
        double per, dec;
        if (GPRN_FILL_FAST) {
            const double s2 = sin_sq_rad(div_rn(PI_D * fabs(r), q[2], aux ? aux[1] : 1.0 / q[2]));
            per = div_rn(-2 * s2, lp2, aux ? aux[0] : 1.0 / lp2);
            dec = div_rn(r * r, le2, aux ? aux[2] : 1.0 / le2);
        } else { const double sn = sin(PI_D * fabs(r) / q[2]); per = -2 * (sn * sn) / lp2; dec = (r * r) / le2; }
        return q[0] * q[0] * (GPRN_FILL_FAST ? exp_neg(per - dec) : exp(per - dec));
    }
    case GPRN_K_RQ:
        return q[0] * q[0] * pow(1 + 0.5 * (r * r) / (q[1] * (q[2] * q[2])), -q[1]);
    case GPRN_K_RQP: {
        const double s = sin(PI_D * fabs(r) / q[3]);
        const double per = exp(-2 * (s * s) / (q[4] * q[4]));
        return q[0] * q[0] * per * pow(1 + (r * r) / (2 * q[1] * (q[2] * q[2])), -q[1]);
    }
    case GPRN_K_COSINE: return q[0] * q[0] * cos(2 * PI_D * fabs(r) / q[1]);
    case GPRN_K_EXPONENTIAL: return q[0] * q[0] * exp(-fabs(r) / q[1]);
    case GPRN_K_MATERN32: {
        const double x = sqrt(3.0) * fabs(r) / q[1];
        return q[0] * q[0] * (1.0 + x) * exp(-x);
    }
    case GPRN_K_MATERN52: {
        const double a = fabs(r), ell = q[1];
        const double poly = 1.0 + (3 * sqrt(5.0) * ell * a + 5 * (a * a)) / (3 * (ell * ell));
        return q[0] * q[0] * poly * exp(-sqrt(5.0) * a / ell);
    }
    case GPRN_K_GAMMAEXP: return q[0] * q[0] * exp(-pow(fabs(r) / q[2], q[1]));
    case GPRN_K_PIECEWISE: {
        const double x = fabs(r / (0.5 * q[0]));
        const double y = 1 - x;
        return x > 1 ? 0.0 : (3 * x + 1) * (y * y * y);
    }
    case GPRN_K_PACIOREK: {
        const double s = q[1] * q[1] + q[2] * q[2];
        return q[0] * q[0] * sqrt(2 * q[1] * q[2] / s) * exp(-2 * r * r / s);
    }
    case GPRN_K_NEWPERIODIC: {
        const double s = sin(PI_D * fabs(r) / q[2]);
        return q[0] * q[0] * pow(1 + 2 * (s * s) / (q[1] * (q[3] * q[3])), -q[1]);
    }
    case GPRN_K_QUASINEWPERIODIC: {
        const double s = sin(PI_D * fabs(r) / q[3]);
        const double a = pow(1 + 2 * (s * s) / (q[1] * (q[4] * q[4])), -q[1]);
        const double b = exp(-0.5 * (r * r) / (q[2] * q[2]));
        return q[0] * q[0] * a * b;
    }
    case GPRN_K_COSPERIODIC: {
        const double c = cos(PI_D * fabs(r) / q[1]);
        return q[0] * q[0] * exp(-2 * (c * c) / (q[2] * q[2]));
    }
    case GPRN_K_QUASICOSPERIODIC: {
        const double c = cos(PI_D * fabs(r) / q[2]);
        return q[0] * q[0] * exp(-2 * (c * c) / (q[3] * q[3]) - (r * r) / (2 * (q[1] * q[1])));
    }
    case GPRN_K_POLYNOMIAL: return pow(q[0] * ti * tj + q[1], q[2]);
    case GPRN_K_HARMONICPERIODIC: {
        double s1, u1, s2, u2;
        harmonic_terms(q[0], q[2], ti, s1, u1);
        harmonic_terms(q[0], q[2], tj, s2, u2);
        const double d2 = (s1 - s2) * (s1 - s2) + (u1 - u2) * (u1 - u2);
        return q[1] * q[1] * exp(-0.5 * d2 / (q[3] * q[3]));
    }
    case GPRN_K_QUASIHARMONICPERIODIC: {
        double s1, u1, s2, u2;
        harmonic_terms(q[0], q[3], ti, s1, u1);
        harmonic_terms(q[0], q[3], tj, s2, u2);
        const double d2 = (s1 - s2) * (s1 - s2) + (u1 - u2) * (u1 - u2);
        const double a = exp(-0.5 * d2 / (q[4] * q[4]));
        const double b = exp(-0.5 * (r * r) / (q[2] * q[2]));
        return q[1] * q[1] * a * b;
    }
    case GPRN_K_DSE: {
        const double e2 = q[1] * q[1];
        return (q[0] * q[0] / (e2 * e2)) * (e2 - r * r) * exp(-0.5 * (r * r) / e2);
    }
    case GPRN_K_DPERIODIC: {
        const double x = PI_D * r / q[1];
        const double sx = sin(x), cx = cos(x);
        const double poly = q[2] * q[2] * cos(2 * x) - 4 * (sx * sx) * (cx * cx);
        return 4 * (PI_D * PI_D) * (q[0] * q[0]) * poly * exp(-2 * (sx * sx) / (q[2] * q[2]));
    }
    case GPRN_K_DQP: {
        const double th = q[0], le = q[1], P = q[2], lp = q[3];
        const double P2 = P * P, lp2 = lp * lp, lp4 = lp2 * lp2, le2 = le * le, le4 = le2 * le2;
        const double sx = sin(PI_D * r / P), cx = cos(PI_D * r / P);
        const double scale = 2 * (th * th) / (P2 * lp4 * le4);
        const double poly = P2 * lp4 * le2 - 2 * P2 * lp4 * (r * r)
            - 4 * PI_D * P * lp2 * le2 * r * sin(2 * PI_D * r / P)
            + 2 * (PI_D * PI_D) * lp2 * le4 * cos(2 * PI_D * r / P)
            - 8 * (PI_D * PI_D) * le4 * (sx * sx) * (cx * cx);
        const double env = exp(-(lp2 * (r * r) + 2 * le2 * (sx * sx)) / (lp2 * le2));
        return scale * poly * env;
    }
    default: return 0.0;
    }
}

__device__ __forceinline__ double eval_program(const FillProgram& pg, double ti, double tj, bool diag)
{
    double st[8];
    int sp = 0;
    for (int o = 0; o < pg.n_ops; ++o) {
        const int op = pg.ops[3 * o];
        if (op == GPRN_OP_PUSH) {
            st[sp & 7] = eval_kernel(pg.ops[3 * o + 1], pg.par + pg.ops[3 * o + 2], ti, tj, diag);
            ++sp;
        } else {
            const double b = st[(sp - 1) & 7], a = st[(sp - 2) & 7];
            st[(sp - 2) & 7] = (op == GPRN_OP_ADD) ? a + b : a * b;
            --sp;
        }
    }
    return st[0];
}

// One instantiation per built-in kernel id (KID >= 0: the switch in eval_kernel folds away, each
// kernel carries only its own registers -- the all-in-one version needed 288 VGPRs, one wave per
// SIMD, and ran 8x slower) plus the generic postfix-program version (KID = -1) for composites.
template <int KID>
__device__ __forceinline__ double eval_any(const FillProgram& pg, double ti, double tj, bool diag)
{
    if constexpr (KID == GPRN_K_SE || KID == GPRN_K_PERIODIC || KID == GPRN_K_QP) return eval_kernel(KID, pg.par, ti, tj, diag, pg.aux);
    else if constexpr (KID >= 0) return eval_kernel(KID, pg.par, ti, tj, diag);
    else return eval_program(pg, ti, tj, diag);
}

// one block = 8 rows x 256 columns; each thread 8 rows x 1 column -> per row the
// 256 threads write 2 KiB contiguous.  Padded region (>= N) becomes identity so
// the blocked factorisation can run on whole tiles.
template <int KID>
__global__ __launch_bounds__(256)
void k_fill(FillProgram pg, const double* __restrict__ t, double* __restrict__ K, int N, int ld,
            const double* __restrict__ diag_add)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 8;
    if (n >= ld) return;
    const double tn = (n < N) ? t[n] : 0.0;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + i;
        if (m >= ld) break;
        double v;
        if (m < N && n < N) {
            v = eval_any<KID>(pg, t[m], tn, m == n);
            if (m == n) {
                if (pg.nugget) v += pg.nugget_val;
                if (diag_add) v += diag_add[m];
            }
        } else {
            v = (m == n) ? 1.0 : 0.0;
        }
        K[(size_t)m * ld + n] = v;
    }
}

// Every built-in but Polynomial is an even function of t_i - t_j evaluated through r*r, |r| or
// sin/cos pairs whose signs cancel, i.e. K is symmetric to the last bit: compute the lower 64-column
// blocks only and write each element twice, the mirror image through an LDS transpose.  Halves the VALU work.
//
// One workgroup = one 64 x 64 block; a thread = TWO adjacent columns x 8 rows: the two evaluations are independent
// instruction streams the compiler interleaves (the exp / sin chains are latency-bound at this occupancy), and both
// images go out as 16-byte stores in 512-byte row segments.  Measured stand-alone on config 3's shapes (N = 4096, ten
// matrices in turn so that the infinity cache does not absorb the writes): SE 24.7 us per matrix = 5.4 TB/s, QP 27.4 us =
// 4.9 TB/s, a kernel that stores a constant in the same pattern 23.6 us = 5.7 TB/s (32-row strips, 256-byte segments in
// the mirror image: 5 % slower); one column per thread with the device library's exp / sinpi, as in rounds 1-2: 29.6 /
// 40.8 us in the same bench.  (Into ONE matrix over and over the same kernels take 20.5 / 23.6 us: the 256 MB cache.)
template <int KID>
__global__ __launch_bounds__(256)
void k_fill_sym(FillProgram pg, const double* __restrict__ t, double* __restrict__ K, int N, int ld,
                const double* __restrict__ diag_add)
{
    constexpr int TR = 64;
    __shared__ double tile[TR][65];
    // lower-triangular block index -> (bi, bj), bi >= bj
    const int L = blockIdx.x, sub = 0;
    int bi = (int)((sqrt(8.0 * L + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= L) ++bi;
    while (bi * (bi + 1) / 2 > L) --bi;
    const int bj = L - bi * (bi + 1) / 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = bj * 64 + 2 * tx;                // ld is a multiple of 128: n + 1 < ld
    const double tn0 = (n < N) ? t[n] : 0.0, tn1 = (n + 1 < N) ? t[n + 1] : 0.0;
    const int row0 = bi * 64 + sub * TR;
#pragma unroll 2
    for (int i = 0; i < TR / 8; ++i) {
        const int r = ty + 8 * i, m = row0 + r;
        const double tm = (m < N) ? t[m] : 0.0;
        // (both evaluated unconditionally, on zeros in the padding: one straight-line instruction stream for the pair)
        double v0 = eval_any<KID>(pg, tm, tn0, m == n);
        double v1 = eval_any<KID>(pg, tm, tn1, m == n + 1);
        if (m == n || m == n + 1) {
            double d = (m == n) ? v0 : v1;
            if (pg.nugget) d += pg.nugget_val;
            if (diag_add && m < N) d += diag_add[m];
            if (m == n) v0 = d; else v1 = d;
        }
        // the padded region (>= N) becomes identity so that the blocked factorisation can run on whole tiles
        if (m >= N || n >= N) v0 = (m == n) ? 1.0 : 0.0;
        if (m >= N || n + 1 >= N) v1 = (m == n + 1) ? 1.0 : 0.0;
        *(double2*)(K + (size_t)m * ld + n) = make_double2(v0, v1);
        tile[r][2 * tx] = v0;
        tile[r][2 * tx + 1] = v1;
    }
    if (bi == bj) return;                          // a diagonal block is complete as computed
    __syncthreads();
    // mirror image: row = a column of the block, TR entries = 32 pairs
    constexpr int CP = TR / 2;
#pragma unroll
    for (int idx = threadIdx.x; idx < 64 * CP; idx += 256) {
        const int c = idx / CP, rp = idx % CP;
        *(double2*)(K + (size_t)(bj * 64 + c) * ld + row0 + 2 * rp) = make_double2(tile[2 * rp][c], tile[2 * rp + 1][c]);
    }
}

static bool program_is_even(const FillProgram& pg)
{
    for (int o = 0; o < pg.n_ops; ++o)
        if (pg.ops[3 * o] == GPRN_OP_PUSH && pg.ops[3 * o + 1] == GPRN_K_POLYNOMIAL) return false;
    return true;
}

// host-side choice of the instantiation: the id of a one-kernel program, else the generic one
#define GPRN_FOR_EACH_KID(X) \
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
    X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
static_assert(GPRN_K_COUNT == 24, "add the new kernel id to GPRN_FOR_EACH_KID");

static int program_kid(const FillProgram& pg)
{
    return (pg.n_ops == 1 && pg.ops[1] >= 0 && pg.ops[1] < GPRN_K_COUNT && pg.ops[2] == 0) ? pg.ops[1] : -1;
}

static void make_program(const KernelSpec& ks, double nugget_val, FillProgram& pg)
{
    pg.n_ops = ks.n_ops;
    pg.nugget = ks.nugget;
    pg.nugget_val = nugget_val;
    for (int i = 0; i < 3 * ks.n_ops; ++i) pg.ops[i] = ks.ops[i];
    for (int i = 0; i < ks.n_params; ++i) pg.par[i] = ks.params[i];
    for (int i = ks.n_params; i < GPRN_MAX_KPARAMS; ++i) pg.par[i] = 0.0;
    for (int i = 3 * ks.n_ops; i < 3 * GPRN_MAX_OPS; ++i) pg.ops[i] = 0;
    pg.aux[0] = pg.aux[1] = pg.aux[2] = 0.0;
    const double* q = pg.par;
    switch (program_kid(pg)) {
    case GPRN_K_SE: pg.aux[0] = 1.0 / (q[1] * q[1]); break;
    case GPRN_K_PERIODIC: pg.aux[0] = 1.0 / (q[2] * q[2]); pg.aux[1] = 1.0 / q[1]; break;
    case GPRN_K_QP: pg.aux[0] = 1.0 / (q[3] * q[3]); pg.aux[1] = 1.0 / q[2]; pg.aux[2] = 1.0 / (2 * (q[1] * q[1])); break;
    default: break;
    }
}

// ---- gradient of the ELBO in the hyper-parameters of ANY kernel program: per parameter l
//   < 1/2 (P - Kinv + a a^T), (K(theta + h e_l) - K(theta - h e_l)) / 2h >,
// the central difference of the program itself (relative step 1e-6, as covFunction._dk_dpars does on the host for
// kernels without a closed form), evaluated and contracted on the fly: one wave per row, rows summed in a fixed order.
// Nothing N x N is written or leaves the GPU.  The nugget is a constant of the parameters and drops out.
__global__ __launch_bounds__(256)
void k_grad_fd_rows(FillProgram pp, FillProgram pm, double inv2h, const double* __restrict__ t,
                    const double* __restrict__ Kinv, const double* __restrict__ P, const double* __restrict__ a,
                    int N, int ld, double* __restrict__ part /* N */)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= N) return;
    const double tm = t[m], am = a[m];
    double acc = 0.0;
    for (int n = lane; n < N; n += 64) {
        const double G = 0.5 * (P[(size_t)m * ld + n] - Kinv[(size_t)m * ld + n] + am * a[n]);
        const double tn = t[n];
        acc += G * (eval_program(pp, tm, tn, m == n) - eval_program(pm, tm, tn, m == n));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) part[m] = acc * inv2h;
}

__global__ __launch_bounds__(256)
void k_sum_fixed(const double* __restrict__ part, int n, double* __restrict__ out)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// out[l], l < ks.n_params (device memory); a = Kinv m already formed; part: N doubles of scratch
int launch_grad_fd(gprn_ctx* c, const KernelSpec& ks, const double* Kinv, const double* P, const double* a,
                   double* part, double* out)
{
    prof_begin(c, GPRN_T_VEC);
    for (int l = 0; l < ks.n_params; ++l) {
        FillProgram pp, pm;
        make_program(ks, 0.0, pp);
        make_program(ks, 0.0, pm);
        const double v = ks.params[l], h = 1e-6 * fmax(1.0, fabs(v));
        pp.par[l] = v + h;
        pm.par[l] = v - h;
        hipLaunchKernelGGL(k_grad_fd_rows, dim3((c->N + 3) / 4), dim3(256), 0, c->stream, pp, pm, 1.0 / (2 * h), c->d_time,
                           Kinv, P, a, c->N, c->ld, part);
        hipLaunchKernelGGL(k_sum_fixed, dim3(1), dim3(256), 0, c->stream, (const double*)part, c->N, out + l);
    }
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

int launch_fill(gprn_ctx* c, const KernelSpec& ks, double* K, double nugget_val, const double* diag_add)
{
    FillProgram pg;
    make_program(ks, nugget_val, pg);
    prof_begin(c, GPRN_T_FILL);
    static int use_sym = -1;                       // GPRN_FILL_SYM=0: always the full-matrix kernel
    if (use_sym < 0) { const char* e = getenv("GPRN_FILL_SYM"); use_sym = e ? atoi(e) : 1; }
    if (use_sym && program_is_even(pg)) {          // ld is a multiple of 128
        const int nb = c->ld / 64;
        dim3 tri(nb * (nb + 1) / 2);
        switch (program_kid(pg)) {
#define X(id) case id: hipLaunchKernelGGL(k_fill_sym<id>, tri, dim3(256), 0, c->stream, pg, c->d_time, K, c->N, c->ld, diag_add); break;
        GPRN_FOR_EACH_KID(X)
#undef X
        default: hipLaunchKernelGGL(k_fill_sym<-1>, tri, dim3(256), 0, c->stream, pg, c->d_time, K, c->N, c->ld, diag_add);
        }
        prof_end(c);
        HIP_TRY(c, hipGetLastError());
        return GPRN_OK;
    }
    dim3 grid((c->ld + 255) / 256, (c->ld + 7) / 8);
    switch (program_kid(pg)) {
#define X(id) case id: hipLaunchKernelGGL(k_fill<id>, grid, dim3(256), 0, c->stream, pg, c->d_time, K, c->N, c->ld, diag_add); break;
    GPRN_FOR_EACH_KID(X)
#undef X
    default: hipLaunchKernelGGL(k_fill<-1>, grid, dim3(256), 0, c->stream, pg, c->d_time, K, c->N, c->ld, diag_add);
    }
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// ---- many matrices in one launch (gprn_elbocalc_batch: smalln.hip, midn.hip): matrix b of the launch has its own program
// (device memory) and its own destination(s); one workgroup per lower 64 x 64 block as in k_fill_sym, the same two columns x
// 8 rows per thread.  The program comes into LDS once per workgroup; the three kernels that carry host-computed reciprocals
// (SE, Periodic, QP) take their handful of parameters into registers and run their own instruction stream, so that a
// matrix filled here has the bits of one filled by launch_fill; everything else goes through the postfix program.
// (The first version read the program from global memory inside the element loop: 1.04 ms for 256 matrices of 512^2,
// 0.5 TB/s -- profiles/r05_batch512_first_kernel_stats.txt.)  K2: a second copy of every matrix (the set-up factors a
// copy of K in place), or null.
template <int KID>
__device__ __forceinline__ void fill_sym_batch_block(const FillProgram& pg, const double* __restrict__ t, double* __restrict__ K,
                                                     double* __restrict__ K2, int N, int ld, double (*tile)[65])
{
    constexpr int TR = 64;
    double par[4] = {0.0, 0.0, 0.0, 0.0}, aux[3] = {0.0, 0.0, 0.0};
    if (KID >= 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) par[i] = pg.par[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) aux[i] = pg.aux[i];
    }
    const int nugget = pg.nugget;
    const double nugget_val = pg.nugget_val;
    const int L = blockIdx.x;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= L) ++bi;
    const int bj = L - bi * (bi + 1) / 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = bj * 64 + 2 * tx;
    const double tn0 = (n < N) ? t[n] : 0.0, tn1 = (n + 1 < N) ? t[n + 1] : 0.0;
    const int row0 = bi * 64;
    auto eval = [&](double ti, double tj, bool diag) {
        if constexpr (KID >= 0) return eval_kernel(KID, par, ti, tj, diag, aux);
        else return eval_program(pg, ti, tj, diag);
    };
#pragma unroll 2
    for (int i = 0; i < TR / 8; ++i) {
        const int r = ty + 8 * i, m = row0 + r;
        const double tm = (m < N) ? t[m] : 0.0;
        double v0 = eval(tm, tn0, m == n);
        double v1 = eval(tm, tn1, m == n + 1);
        if (m == n || m == n + 1) {
            double d = (m == n) ? v0 : v1;
            if (nugget) d += nugget_val;
            if (m == n) v0 = d; else v1 = d;
        }
        if (m >= N || n >= N) v0 = (m == n) ? 1.0 : 0.0;
        if (m >= N || n + 1 >= N) v1 = (m == n + 1) ? 1.0 : 0.0;
        *(double2*)(K + (size_t)m * ld + n) = make_double2(v0, v1);
        if (K2) *(double2*)(K2 + (size_t)m * ld + n) = make_double2(v0, v1);
        tile[r][2 * tx] = v0;
        tile[r][2 * tx + 1] = v1;
    }
    if (bi == bj) return;
    __syncthreads();
    constexpr int CP = TR / 2;
    for (int idx = threadIdx.x; idx < 64 * CP; idx += 256) {
        const int c = idx / CP, rp = idx % CP;
        const double2 v = make_double2(tile[2 * rp][c], tile[2 * rp + 1][c]);
        *(double2*)(K + (size_t)(bj * 64 + c) * ld + row0 + 2 * rp) = v;
        if (K2) *(double2*)(K2 + (size_t)(bj * 64 + c) * ld + row0 + 2 * rp) = v;
    }
}

__global__ __launch_bounds__(256)
void k_fill_sym_batch(const FillProgram* __restrict__ pgs, const double* __restrict__ t, double* const* __restrict__ Ks,
                      double* const* __restrict__ K2s, int N, int ld)
{
    __shared__ double tile[64][65];
    __shared__ FillProgram spg;
    {
        const int* src = reinterpret_cast<const int*>(pgs + blockIdx.y);
        int* dst = reinterpret_cast<int*>(&spg);
        for (int i = threadIdx.x; i < (int)(sizeof(FillProgram) / sizeof(int)); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    double* const K = Ks[blockIdx.y];
    double* const K2 = K2s ? K2s[blockIdx.y] : nullptr;
    // (uniform per workgroup: a scalar branch)
    const int kid = (spg.n_ops == 1 && spg.ops[0] == GPRN_OP_PUSH && spg.ops[2] == 0) ? spg.ops[1] : -1;
    switch (kid) {
    case GPRN_K_SE: fill_sym_batch_block<GPRN_K_SE>(spg, t, K, K2, N, ld, tile); break;
    case GPRN_K_PERIODIC: fill_sym_batch_block<GPRN_K_PERIODIC>(spg, t, K, K2, N, ld, tile); break;
    case GPRN_K_QP: fill_sym_batch_block<GPRN_K_QP>(spg, t, K, K2, N, ld, tile); break;
    default: fill_sym_batch_block<-1>(spg, t, K, K2, N, ld, tile);
    }
}

size_t fill_program_bytes() { return sizeof(FillProgram); }

// the program of `ks` (1e-6 nugget for one-argument kernels: meanfield.py:433) with other parameter values, written to dst;
// false when the program is not an even function of t_i - t_j (Polynomial: the symmetric fill does not apply)
bool fill_program_with(const KernelSpec& ks, const double* params, void* dst)
{
    KernelSpec k2 = ks;
    for (int i = 0; i < ks.n_params; ++i) k2.params[i] = params[i];
    FillProgram pg;
    make_program(k2, 1e-6, pg);
    memcpy(dst, &pg, sizeof(pg));
    return program_is_even(pg);
}

// n_matrices matrices of the context's N (ld = 128 T) from d_programs[i] into d_Ks[i]
int launch_fill_batch(gprn_ctx* c, const void* d_programs, double* const* d_Ks, int n_matrices, double* const* d_K2s)
{
    if (n_matrices <= 0) return GPRN_OK;
    prof_begin(c, GPRN_T_FILL);
    const int nb = c->ld / 64;
    hipLaunchKernelGGL(k_fill_sym_batch, dim3(nb * (nb + 1) / 2, n_matrices), dim3(256), 0, c->stream,
                       (const FillProgram*)d_programs, c->d_time, d_Ks, d_K2s, c->N, c->ld);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}

// Rectangular cross-covariance K*[i][n] = k(t*_i, t_n), no nugget (_gp.py:50-61, meanfield.py:455-471),
// rows padded to a multiple of 128 with zeros; and the prior variance at the prediction points,
// kss[i] = k(t*_i, t*_i) + nugget (the diagonal of _gp.py:40-48 evaluated at tstar).
template <int KID>
__global__ __launch_bounds__(256)
void k_fill_rect(FillProgram pg, const double* __restrict__ ts, int ns, int ns_pad,
                 const double* __restrict__ t, int N, int ld, double* __restrict__ Ks,
                 double* __restrict__ kss)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 8;
    if (n >= ld) return;
    const double tn = (n < N) ? t[n] : 0.0;
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        const int i = i0 + r;
        if (i >= ns_pad) break;
        double v = 0.0;
        if (i < ns && n < N)
            v = eval_any<KID>(pg, ts[i], tn, false);
        Ks[(size_t)i * ld + n] = v;
        if (n == 0 && i < ns) {
            double d = eval_any<KID>(pg, ts[i], ts[i], true);
            if (pg.nugget) d += pg.nugget_val;
            kss[i] = d;
        }
    }
}

int launch_fill_rect(gprn_ctx* c, const KernelSpec& ks, double nugget_val, const double* d_tstar,
                     int ns, int ns_pad, double* Ks, double* kss)
{
    FillProgram pg;
    make_program(ks, nugget_val, pg);
    prof_begin(c, GPRN_T_FILL);
    dim3 grid((c->ld + 255) / 256, (ns_pad + 7) / 8);
    switch (program_kid(pg)) {
#define X(id) case id: hipLaunchKernelGGL(k_fill_rect<id>, grid, dim3(256), 0, c->stream, pg, d_tstar, ns, ns_pad, c->d_time, c->N, c->ld, Ks, kss); break;
    GPRN_FOR_EACH_KID(X)
#undef X
    default: hipLaunchKernelGGL(k_fill_rect<-1>, grid, dim3(256), 0, c->stream, pg, d_tstar, ns, ns_pad, c->d_time, c->N, c->ld, Ks, kss);
    }
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    return GPRN_OK;
}
