"""ctypes binding of libgprn_hip.so (C ABI: include/gprn_hip.h).

Thin by design: NumPy arrays in, NumPy arrays out, status codes turned into
exceptions.  There is no CPU fallback anywhere in this package -- if the
library or a GPU is missing, `Context()` raises `BackendUnavailable`.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GPRN_HIP_LIB') or os.path.join(_HERE, 'libgprn_hip.so')

GPRN_E_ARG, GPRN_E_HIP, GPRN_E_NODEV, GPRN_E_COMM, GPRN_E_NOMEM, GPRN_E_UNSUPPORTED = -1, -2, -3, -4, -5, -6
M_K, M_KLINV, M_SIGMA, M_BX, M_BL = 0, 1, 2, 3, 4
T_NAMES = ('fill', 'build_B', 'diag', 'panel', 'update', 'lauum', 'vec', 'update_ahead')
TILE = 128

_dp = POINTER(c_double)


class BackendUnavailable(RuntimeError):
    """libgprn_hip.so is not built, or there is no MI355X to run it on."""


class BackendError(RuntimeError):
    pass


# every exported entry point: name -> (restype, argtypes).  tests/test_abi.py
# checks this table against include/gprn_hip.h and the built library.
SIGNATURES = {
    'gprn_device_count': (c_int, []),
    'gprn_create': (c_int, [POINTER(c_void_p), c_int]),
    'gprn_destroy': (None, [c_void_p]),
    'gprn_last_error': (c_char_p, [c_void_p]),
    'gprn_last_info_gp': (c_int, [c_void_p]),
    'gprn_set_data': (c_int, [c_void_p, c_int, c_int, c_int, _dp, _dp, _dp]),
    'gprn_comm_unique_id': (c_int, [POINTER(c_char)]),
    'gprn_comm_init': (c_int, [c_void_p, c_int, c_int, POINTER(c_char)]),
    'gprn_set_owners': (c_int, [c_void_p, POINTER(c_int)]),
    'gprn_comm_barrier_max': (c_int, [c_void_p, _dp]),
    'gprn_set_kernel': (c_int, [c_void_p, c_int, POINTER(c_int32), c_int, _dp, c_int, c_int]),
    'gprn_upload_K': (c_int, [c_void_p, c_int, _dp]),
    'gprn_set_y_resid': (c_int, [c_void_p, _dp]),
    'gprn_set_jitters': (c_int, [c_void_p, _dp]),
    'gprn_factor_priors': (c_int, [c_void_p]),
    'gprn_set_muvar': (c_int, [c_void_p, _dp, _dp]),
    'gprn_get_muvar': (c_int, [c_void_p, _dp, _dp]),
    'gprn_sweep': (c_int, [c_void_p, c_int, c_int, _dp, _dp]),
    'gprn_predict': (c_int, [c_void_p, c_int, _dp, _dp, _dp]),
    'gprn_predict_upload': (c_int, [c_void_p, c_int, c_int, _dp, _dp, _dp]),
    'gprn_get_scalars': (c_int, [c_void_p, _dp]),
    'gprn_keep_sigma': (c_int, [c_void_p, c_int]),
    'gprn_get_matrix': (c_int, [c_void_p, c_int, c_int, _dp]),
    'gprn_get_logdet_K': (c_int, [c_void_p, _dp]),
    'gprn_profile_enable': (c_int, [c_void_p, c_int]),
    'gprn_profile_read': (c_int, [c_void_p, _dp, POINTER(c_int64), c_int]),
    'gprn_test_gemm': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, _dp, _dp, _dp]),
    'gprn_comm_allreduce_sum': (c_int, [c_void_p, _dp, c_int]),
    'gprn_test_factor_invert': (c_int, [c_void_p, c_int, c_int, _dp, _dp, _dp]),
    'gprn_test_lauum': (c_int, [c_void_p, c_int, _dp, _dp]),
    'gprn_test_gemm_rate': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, _dp]),
    'gprn_test_fill_rate': (c_int, [c_void_p, c_int, _dp]),
    'gprn_test_mfma_peak': (c_int, [c_void_p, c_int, c_int, _dp]),
    'gprn_set_option': (c_int, [c_void_p, c_char_p, c_int, POINTER(c_int)]),
    'gprn_elbocalc_batch': (c_int, [c_void_p, c_int, _dp, c_int, _dp, _dp, _dp, _dp, c_int, _dp, POINTER(c_int), POINTER(c_int),
                            POINTER(c_int), _dp, _dp]),
    'gprn_elbocalc': (c_int, [c_void_p, c_int, _dp, _dp, _dp, _dp, c_int, _dp, c_int, POINTER(c_int), POINTER(c_int),
                      POINTER(c_int), _dp, _dp]),
    'gprn_expected_loglike': (c_int, [c_void_p, _dp]),
    'gprn_prior_terms': (c_int, [c_void_p, c_int, _dp, _dp, _dp]),
    'gprn_grad_matrices': (c_int, [c_void_p, c_int, _dp, _dp]),
    'gprn_grad_kernel': (c_int, [c_void_p, c_int, _dp, _dp]),
    'gprn_eval_kernel': (c_int, [c_void_p, POINTER(c_int32), c_int, _dp, c_int, c_double, _dp]),
    'gprn_sample_prior': (c_int, [c_void_p, POINTER(c_int32), c_int, _dp, c_int, c_double, c_int, _dp, _dp]),
}

_lib = None


def load_library():
    """dlopen the in-tree library and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; '
            f'g.build()"` (or make -C gpyrn_amd/csrc)')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise BackendUnavailable(f'cannot load {LIB_PATH}: {exc}') from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count():
    return int(load_library().gprn_device_count())


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f'expected shape {tuple(shape)}, got {a.shape}')
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp)


def comm_unique_id():
    buf = ctypes.create_string_buffer(128)
    rc = load_library().gprn_comm_unique_id(buf)
    if rc:
        raise BackendError(f'gprn_comm_unique_id failed ({rc}): is librccl present?')
    return buf.raw


class Context:
    """One GPU.  Mirrors the C handle; methods map 1:1 onto the C entry points."""

    def __init__(self, device=0):
        self._h = None
        self._lib = load_library()
        h = c_void_p()
        rc = self._lib.gprn_create(byref(h), int(device))
        if rc == GPRN_E_NODEV:
            raise BackendUnavailable('no HIP device visible: gpyrn_amd has no CPU path '
                                     '(the NumPy oracle lives in oracle/, for tests only)')
        if rc:
            raise BackendError(f'gprn_create({device}) failed with {rc}')
        self._h = h
        self.device = int(device)
        self.N = self.p = self.q = self.G = 0
        self.world, self.rank = 1, 0

    def close(self):
        if self._h is not None:
            self._lib.gprn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- status ------------------------------------------------------------
    def _check(self, rc, what):
        """Negative codes raise; a positive code is a LAPACK-style info and is returned."""
        if rc < 0:
            msg = self._lib.gprn_last_error(self._h)
            raise BackendError(f'{what}: {msg.decode() if msg else rc} (code {rc})')
        return rc

    @property
    def last_info_gp(self):
        return int(self._lib.gprn_last_info_gp(self._h))

    # -- problem -----------------------------------------------------------
    def set_data(self, time, y, yerr, q):
        time = _f64(time)
        y = _f64(y)
        yerr = _f64(yerr, y.shape)
        p, N = y.shape
        if time.shape != (N,):
            raise ValueError('time and y disagree')
        self._check(self._lib.gprn_set_data(self._h, N, p, int(q), _ptr(time), _ptr(y), _ptr(yerr)),
                    'set_data')
        self.N, self.p, self.q, self.G = N, p, int(q), int(q) * (p + 1)

    def comm_init(self, world, rank, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128) if unique_id else None
        self._check(self._lib.gprn_comm_init(self._h, int(world), int(rank), buf), 'comm_init')
        self.world, self.rank = int(world), int(rank)

    def set_owners(self, owner):
        arr = (c_int * self.G)(*[int(o) for o in owner])
        self._check(self._lib.gprn_set_owners(self._h, arr), 'set_owners')
        self._owner = [int(o) for o in owner]

    def owner_of(self, gp):
        """Rank that factors latent GP `gp` (0 on an unsharded context)."""
        owner = getattr(self, '_owner', None)
        return owner[gp] if owner else 0

    def barrier_max(self, value=0.0):
        v = c_double(float(value))
        self._check(self._lib.gprn_comm_barrier_max(self._h, byref(v)), 'barrier_max')
        return v.value

    def allreduce_sum(self, values):
        """Sum of a float64 vector over the ranks of this context's communicator."""
        buf = np.array(values, dtype=np.float64).ravel()
        self._check(self._lib.gprn_comm_allreduce_sum(self._h, _ptr(buf), buf.size), 'allreduce_sum')
        return buf

    def set_kernel(self, gp, ops, params, add_nugget):
        flat = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 3))
        par = _f64(np.atleast_1d(params))
        self._check(self._lib.gprn_set_kernel(
            self._h, int(gp), flat.ctypes.data_as(POINTER(c_int32)), flat.shape[0],
            _ptr(par), par.size, int(bool(add_nugget))), 'set_kernel')

    def upload_K(self, gp, K):
        K = _f64(K, (self.N, self.N))
        self._check(self._lib.gprn_upload_K(self._h, int(gp), _ptr(K)), 'upload_K')

    def set_y_resid(self, y):
        y = _f64(y, (self.p, self.N))
        self._check(self._lib.gprn_set_y_resid(self._h, _ptr(y)), 'set_y_resid')

    def set_jitters(self, jitters):
        j = _f64(np.atleast_1d(jitters), (self.p,))
        self._check(self._lib.gprn_set_jitters(self._h, _ptr(j)), 'set_jitters')

    def factor_priors(self):
        return self._check(self._lib.gprn_factor_priors(self._h), 'factor_priors')

    def set_muvar(self, mu, var):
        d = self.N * self.q * (self.p + 1)
        mu = _f64(np.ravel(mu), (d,))
        var = _f64(np.ravel(var), (d,))
        self._check(self._lib.gprn_set_muvar(self._h, _ptr(mu), _ptr(var)), 'set_muvar')

    def get_muvar(self):
        shape = (self.p + 1, self.q, self.N)
        mu, var = np.empty(shape), np.empty(shape)
        self._check(self._lib.gprn_get_muvar(self._h, _ptr(mu), _ptr(var)), 'get_muvar')
        return mu, var

    def sweep(self, n=1, commit=True):
        """n x ELBOaux.  Returns (elbo[n], parts[n,3] = LogL, LogP, Ent, info)."""
        elbo = np.empty(n)
        parts = np.empty((n, 3))
        info = self._check(self._lib.gprn_sweep(self._h, int(n), int(bool(commit)),
                                                _ptr(elbo), _ptr(parts)), 'sweep')
        return elbo, parts, info

    def predict(self, tstar):
        """Conditional mean / variance of every latent GP at `tstar`: two (G, n*) arrays (complete on
        every rank of a sharded context) and the LAPACK-style info."""
        ts = _f64(np.ravel(tstar))
        mean = np.zeros((self.G, ts.size))
        var = np.zeros((self.G, ts.size))
        info = self._check(self._lib.gprn_predict(self._h, ts.size, _ptr(ts), _ptr(mean), _ptr(var)),
                           'predict')
        return mean, var, info

    def predict_upload(self, gp, K_tiny, Kstar, kss):
        """Host-evaluated K + 1.25e-12 I (N, N), K* (n*, N) and k** (n*) of latent GP `gp` for the next predict()."""
        K_tiny = _f64(K_tiny, (self.N, self.N))
        Kstar = _f64(np.atleast_2d(Kstar))
        if Kstar.shape[1] != self.N:
            raise ValueError('Kstar must have N columns')
        kss = _f64(np.ravel(kss), (Kstar.shape[0],))
        self._check(self._lib.gprn_predict_upload(self._h, int(gp), Kstar.shape[0], _ptr(K_tiny), _ptr(Kstar),
                                                  _ptr(kss)), 'predict_upload')

    def get_scalars(self):
        """Per-GP scalars of the last sweep: dict of log det B (G), tr B^-1 (G), m^T K^-1 m (G), Q1 traces (q, q)."""
        out = np.empty(3 * self.G + self.q * self.q)
        self._check(self._lib.gprn_get_scalars(self._h, _ptr(out)), 'get_scalars')
        G = self.G
        return {'logdetB': out[:G].copy(), 'trBinv': out[G:2 * G].copy(), 'muKmu': out[2 * G:3 * G].copy(),
                'q1': out[3 * G:].reshape(self.q, self.q).copy()}

    def eval_kernel(self, ops, params, nugget):
        """K = expr(t_i, t_j) + nugget I at the data times, filled on the device."""
        flat = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 3))
        par = _f64(np.atleast_1d(params))
        K = np.empty((self.N, self.N))
        self._check(self._lib.gprn_eval_kernel(self._h, flat.ctypes.data_as(POINTER(c_int32)), flat.shape[0],
                                               _ptr(par), par.size, float(nugget), _ptr(K)), 'eval_kernel')
        return K

    def sample_prior(self, ops, params, nugget, z):
        """L z for every row z of standard normals, K + nugget I = L L^T; returns (samples, info)."""
        flat = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 3))
        par = _f64(np.atleast_1d(params))
        z = _f64(np.atleast_2d(z))
        if z.shape[1] != self.N:
            raise ValueError('z must have N columns')
        out = np.empty_like(z)
        info = self._check(self._lib.gprn_sample_prior(
            self._h, flat.ctypes.data_as(POINTER(c_int32)), flat.shape[0], _ptr(par), par.size, float(nugget),
            z.shape[0], _ptr(z), _ptr(out)), 'sample_prior')
        return out, info

    def expected_loglike(self):
        """inference._expectedLogLike of the state and jitters last set (gprn_expected_loglike)."""
        v = c_double(0.0)
        self._check(self._lib.gprn_expected_loglike(self._h, byref(v)), 'expected_loglike')
        return v.value

    def prior_terms(self, gp, S, m):
        """(log det K_gp, m^T K_gp^-1 m, tr(K_gp^-1 S)) from the resident factor of K_gp (gprn_prior_terms)."""
        S = _f64(S, (self.N, self.N))
        m = _f64(np.ravel(m), (self.N,))
        out = np.zeros(3)
        self._check(self._lib.gprn_prior_terms(self._h, int(gp), _ptr(S), _ptr(m), _ptr(out)), 'prior_terms')
        return out

    def grad_matrices(self, gp):
        """(K^-1, K^-1 S K^-1) of latent GP `gp` after a sweep with keep_sigma: the N^3 part of the ELBO gradient."""
        Kinv, P = np.empty((self.N, self.N)), np.empty((self.N, self.N))
        self._check(self._lib.gprn_grad_matrices(self._h, int(gp), _ptr(Kinv), _ptr(P)), 'grad_matrices')
        return Kinv, P

    def grad_kernel(self, gp, m, n_params):
        """Kernel-parameter gradient of latent GP `gp` contracted on the device (closed forms for SE / Periodic /
        QuasiPeriodic, central differences of the kernel program otherwise); None for a latent GP whose matrix
        was uploaded (user-defined kernels) -- use grad_matrices then."""
        m = _f64(np.ravel(m), (self.N,))
        out = np.zeros(max(4, int(n_params)))
        rc = self._lib.gprn_grad_kernel(self._h, int(gp), _ptr(m), _ptr(out))
        if rc == GPRN_E_UNSUPPORTED:           # every other error (sharded context, no Sigma kept, ...) raises
            return None
        self._check(rc, 'grad_kernel')
        return out[:n_params]

    def keep_sigma(self, on=True):
        self._check(self._lib.gprn_keep_sigma(self._h, int(bool(on))), 'keep_sigma')

    def get_matrix(self, which, gp):
        out = np.empty((self.N, self.N))
        self._check(self._lib.gprn_get_matrix(self._h, int(which), int(gp), _ptr(out)), 'get_matrix')
        return out

    def get_logdet_K(self):
        out = np.empty(self.G)
        self._check(self._lib.gprn_get_logdet_K(self._h, _ptr(out)), 'get_logdet_K')
        return out

    def elbocalc(self, max_iter, setup=False, y_resid=None, jitters=None, mu=None, var=None):
        """ELBOcalc from its set-up block on (meanfield.py:618-649) in one call of the library: optionally the set-up
        with the kernels last sent, y - mean, the jitters and the starting state (None: as set before), then the loop.
        Returns (elboArray, iterNumber, converged, info, mu, var) with the state the loop ended in."""
        cap = int(min(max_iter, 1 << 20)) + 1
        hist = np.empty(cap)
        d = (self.p + 1) * self.q * self.N
        mu_out, var_out = np.empty(d), np.empty(d)
        n, it, conv = c_int(0), c_int(0), c_int(0)
        opt = lambda a, size: None if a is None else _ptr(_f64(np.ravel(a), (size,)))
        keep = [None if a is None else _f64(np.ravel(a)) for a in (y_resid, jitters, mu, var)]
        ptrs = [None if a is None else _ptr(a) for a in keep]
        if (keep[0] is not None and keep[0].size != self.p * self.N) or (keep[1] is not None and keep[1].size != self.p) or \
                (keep[2] is not None and (keep[2].size != d or keep[3] is None or keep[3].size != d)):
            raise ValueError('elbocalc: y_resid (p, N), jitters (p,), mu and var (d,) expected')
        info = self._check(self._lib.gprn_elbocalc(self._h, 1 if setup else 0, ptrs[0], ptrs[1], ptrs[2], ptrs[3],
                                                   int(max_iter), _ptr(hist), cap, byref(n), byref(it), byref(conv),
                                                   _ptr(mu_out), _ptr(var_out)), 'elbocalc')
        shape = (self.p + 1, self.q, self.N)
        return (hist[:min(n.value, cap)].copy(), it.value, bool(conv.value), info,
                mu_out.reshape(shape), var_out.reshape(shape))

    def elbocalc_batch(self, kernel_params, y_resid, jitters, mu, var, max_iter, want_state=False):
        """B independent ELBOcalc loops side by side (gprn_elbocalc_batch): kernel_params (B, n_kpar), y_resid (B, p, N),
        jitters (B, p), mu / var (B, d).  Returns (elbo[B], iterations[B], converged[B], info[B]) and, with want_state,
        the final states (B, p+1, q, N) twice -- or None where the library has no batched form for this problem."""
        kp = _f64(np.atleast_2d(kernel_params))
        B = kp.shape[0]
        d = (self.p + 1) * self.q * self.N
        yr = _f64(np.reshape(y_resid, (B, self.p * self.N)))
        jt = _f64(np.reshape(jitters, (B, self.p)))
        m0, v0 = _f64(np.reshape(mu, (B, d))), _f64(np.reshape(var, (B, d)))
        elbo = np.empty(B)
        it, cv, info = (np.zeros(B, dtype=np.int32) for _ in range(3))
        mo = np.empty((B, d)) if want_state else None
        vo = np.empty((B, d)) if want_state else None
        ip = lambda a: a.ctypes.data_as(POINTER(c_int))
        rc = self._lib.gprn_elbocalc_batch(self._h, B, _ptr(kp), kp.shape[1], _ptr(yr), _ptr(jt), _ptr(m0), _ptr(v0),
                                           int(max_iter), _ptr(elbo), ip(it), ip(cv), ip(info),
                                           _ptr(mo) if want_state else None, _ptr(vo) if want_state else None)
        if rc == GPRN_E_UNSUPPORTED:
            return None
        self._check(rc, 'elbocalc_batch')
        out = (elbo, it.astype(int), cv.astype(bool), info.astype(int))
        if want_state:
            shape = (B, self.p + 1, self.q, self.N)
            out += (mo.reshape(shape), vo.reshape(shape))
        return out

    def option(self, name, value=-1):
        """Read (value < 0) or set a per-context switch of the library; returns the previous value."""
        old = c_int(0)
        self._check(self._lib.gprn_set_option(self._h, name.encode(), int(value), byref(old)), 'set_option')
        return old.value

    # -- timing --------------------------------------------------------------
    def profile_enable(self, families=T_NAMES):
        mask = 0
        for f in families or ():
            mask |= 1 << T_NAMES.index(f)
        self._check(self._lib.gprn_profile_enable(self._h, mask), 'profile_enable')

    def profile_read(self, reset=True):
        ms = np.zeros(len(T_NAMES))
        n = np.zeros(len(T_NAMES), dtype=np.int64)
        self._check(self._lib.gprn_profile_read(self._h, _ptr(ms), n.ctypes.data_as(POINTER(c_int64)),
                                                int(bool(reset))), 'profile_read')
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(T_NAMES)}

    # -- diagnostics -----------------------------------------------------------
    def test_gemm(self, A, B, C, a_mode, b_mode, c_mode):
        A, B = _f64(A), _f64(B)
        C = _f64(C).copy()
        M, K = A.shape
        K2, N = B.shape
        assert K == K2 and C.shape == (M, N)
        self._check(self._lib.gprn_test_gemm(self._h, M, N, K, a_mode, b_mode, c_mode,
                                             _ptr(A), _ptr(B), _ptr(C)), 'test_gemm')
        return C

    def test_factor_invert(self, A):
        A = _f64(A)
        if A.ndim == 2:
            A = A[None]
        batch, n, _ = A.shape
        L, X = np.empty_like(A), np.empty_like(A)
        info = self._check(self._lib.gprn_test_factor_invert(self._h, n, batch, _ptr(A), _ptr(L), _ptr(X)),
                           'test_factor_invert')
        return L, X, info

    def gemm_rate(self, M, N, K, how, reps=3):
        """TFLOP/s of C -= A.B^T (M x N x K) through the tile contraction: one launch of 64 x 64 (how 0) / 128 x 128 (how 1)
        workgroups."""
        v = c_double(0.0)
        self._check(self._lib.gprn_test_gemm_rate(self._h, int(M), int(N), int(K), int(how), int(reps), byref(v)), 'test_gemm_rate')
        return 2.0 * M * N * K / (v.value * 1e-3) / 1e12

    def fill_rate(self, reps=20):
        """ms per pass over all local covariance fills (the kernels last sent), timed inside one pair of events."""
        v = c_double(0.0)
        self._check(self._lib.gprn_test_fill_rate(self._h, int(reps), byref(v)), 'test_fill_rate')
        return v.value

    def mfma_peak(self, wg_per_cu=1, iters=2000):
        """Measured fp64 MFMA issue ceiling (TFLOP/s) of this device."""
        v = c_double(0.0)
        self._check(self._lib.gprn_test_mfma_peak(self._h, int(wg_per_cu), int(iters), byref(v)),
                    'test_mfma_peak')
        return v.value

    def test_lauum(self, X):
        X = _f64(X)
        out = np.empty_like(X)
        self._check(self._lib.gprn_test_lauum(self._h, X.shape[0], _ptr(X), _ptr(out)), 'test_lauum')
        return out
