"""Covariance functions (kernels) for GPRN nodes and weights.

Plugin surface of the reference's ``gpyrn/covfunc.py`` (base protocol :5-53,
operators :56-104, concrete kernels :107-689), kept name for name so user
scripts and user subclasses keep working:

* ``covFunction(*pars)`` stores ``self.pars`` (float64); ``kernel(r)`` evaluates
  on an array of time differences; ``get_parameters`` / ``set_parameters`` chain
  through a flat vector (``set_parameters`` returns the unconsumed tail);
  ``+`` and ``*`` build ``Sum`` / ``Multiplication``.

What is new here is the *device description*: every built-in kernel can emit a
tiny postfix program (``_device_program``) that the fused HIP covariance-fill
kernel (csrc/fill.hip, ``gprn_set_kernel`` in include/gprn_hip.h) evaluates per
matrix element straight from the time vector, so no N x N ``r`` matrix is ever
materialised for them.  ``kernel(r)`` on the host stays available -- it *is* the
plugin API -- and is what an unknown user subclass is evaluated with before its
matrix is uploaded (``gprn_upload_K``).

Reference behaviours that are kept on purpose, because results must match:
``Sum``/``Multiplication`` evaluate their children with the children's own
``pars`` (a composite's ``set_parameters`` does not reach them, covfunc.py:56-62);
the "attribute" kernels (``Paciorek`` ... ``QuasiCosPeriodic``) evaluate from the
constructor attributes, not from ``pars`` (covfunc.py:493-496,517-519,543-546,
664-665,687-689); ``HarmonicPeriodic`` keeps its ``x / 2*sin`` precedence
(covfunc.py:599-605); ``NewRQP`` fails at call time as the reference does (:574).
"""
import numpy as np

from ._utils import _array_input, _take_leading

__all__ = [
    'covFunction', 'Sum', 'Multiplication', 'Derivative', 'Constant',
    'WhiteNoise', 'SquaredExponential', 'Periodic', 'QuasiPeriodic',
    'RationalQuadratic', 'RQP', 'Cosine', 'Exponential', 'Matern32',
    'Matern52', 'Linear', 'GammaExp', 'Polynomial', 'Piecewise', 'Paciorek',
    'NewPeriodic', 'QuasiNewPeriodic', 'NewRQP', 'HarmonicPeriodic',
    'QuasiHarmonicPeriodic', 'CosPeriodic', 'QuasiCosPeriodic',
]

# Kernel ids understood by csrc/fill.hip -- keep in step with include/gprn_hip.h.
KID = dict(
    CONSTANT=0, WHITENOISE=1, SE=2, PERIODIC=3, QP=4, RQ=5, RQP=6, COSINE=7,
    EXPONENTIAL=8, MATERN32=9, MATERN52=10, GAMMAEXP=11, PIECEWISE=12,
    PACIOREK=13, NEWPERIODIC=14, QUASINEWPERIODIC=15, COSPERIODIC=16,
    QUASICOSPERIODIC=17, POLYNOMIAL=18, HARMONICPERIODIC=19,
    QUASIHARMONICPERIODIC=20, DSE=21, DPERIODIC=22, DQP=23,
)
OP_PUSH, OP_ADD, OP_MUL = 0, 1, 2
# kernels that _KMatrix calls as kernel(t_i, t_j) and leaves without nugget
# (meanfield.py:426-431)
_TWO_ARGUMENT_IDS = (KID['POLYNOMIAL'], KID['HARMONICPERIODIC'],
                     KID['QUASIHARMONICPERIODIC'])


class covFunction:
    """Base class of all kernels (covfunc.py:5-53)."""
    _device_id = None          # built-ins set a KID; user subclasses leave None

    def __init__(self, *args):
        self.pars = np.array(args, dtype=float)

    def __call__(self, r, t1=None, t2=None):
        raise NotImplementedError

    def _dkdxidj(self, r):
        raise NotImplementedError

    def __repr__(self):
        names = getattr(self, '_param_names', None)
        if names is None:
            inner = ', '.join(str(v) for v in self.pars)
        else:
            inner = ', '.join(f'{n}={v}' for n, v in zip(names, self.pars))
        return f'{type(self).__name__}({inner})'

    # -- parameter plumbing -------------------------------------------------
    def get_parameters(self):
        return self.pars

    @_array_input
    def set_parameters(self, p):
        return _take_leading(self, p, 'kernel')

    # -- derivatives in the hyper-parameters (not in the reference: SURVEY.md 8f-3) ----------
    def _dk_dpars(self, r):
        """``[dK/dpars[0], dK/dpars[1], ...]`` on the array of time differences `r`.  The base version
        differentiates ``self(r)`` numerically (central differences, relative step 1e-6), which serves
        every kernel, user subclasses included; built-ins with a closed form override it."""
        out = []
        keep = self.pars.copy()
        try:
            for i, v in enumerate(keep):
                h = 1e-6 * max(1.0, abs(v))
                self.pars = keep.copy(); self.pars[i] = v + h
                up = np.asarray(self(r), dtype=float)
                self.pars = keep.copy(); self.pars[i] = v - h
                dn = np.asarray(self(r), dtype=float)
                out.append((up - dn) / (2 * h))
        finally:
            self.pars = keep
        return out

    # -- algebra -------------------------------------------------------------
    def __add__(self, other):
        return Sum(self, other)

    __radd__ = __add__

    def __mul__(self, other):
        return Multiplication(self, other)

    __rmul__ = __mul__

    # -- device description --------------------------------------------------
    def _device_pars(self):
        """The numbers ``__call__`` would use, in the order fill.hip expects."""
        return self.pars

    def _device_program(self):
        """Postfix program ``(ops, params)`` for the fused fill kernel, or None
        when this kernel has to be evaluated on the host and uploaded."""
        if type(self)._device_id is None or not _is_builtin(type(self)):
            return None
        pars = np.asarray(self._device_pars(), dtype=float).ravel()
        return [(OP_PUSH, type(self)._device_id, 0)], pars


def _is_builtin(cls):
    # a user subclass of a built-in may override __call__: never trust its id
    return cls.__module__ == __name__


class _operator(covFunction):
    """Binary node of a kernel expression (covfunc.py:56-62)."""
    _opcode = None

    def __init__(self, k1, k2):
        self.k1, self.k2 = k1, k2
        self.kerneltype = 'complex'
        self.pars = np.r_[k1.pars, k2.pars]

    def _device_program(self):
        if not _is_builtin(type(self)):
            return None
        left = self.k1._device_program() if isinstance(self.k1, covFunction) else None
        right = self.k2._device_program() if isinstance(self.k2, covFunction) else None
        if left is None or right is None:
            return None
        ops_l, par_l = left
        ops_r, par_r = right
        if any(op == OP_PUSH and kid in _TWO_ARGUMENT_IDS
               for op, kid, _ in ops_l + ops_r):
            return None        # kernel(r) would raise on the host, let it
        shifted = [(op, kid, off + par_l.size if op == OP_PUSH else 0)
                   for op, kid, off in ops_r]
        return ops_l + shifted + [(self._opcode, 0, 0)], np.r_[par_l, par_r]


class Sum(_operator):
    """k1 + k2 (covfunc.py:65-71)."""
    _opcode = OP_ADD

    def __call__(self, r):
        return self.k1(r) + self.k2(r)

    def _dk_dpars(self, r):
        # (the node's own `pars` is a copy of the children's, as in the reference: differentiate the children)
        return list(self.k1._dk_dpars(r)) + list(self.k2._dk_dpars(r))

    def __repr__(self):
        return f'{self.k1} + {self.k2}'


class Multiplication(_operator):
    """k1 * k2 (covfunc.py:74-80)."""
    _opcode = OP_MUL

    def __call__(self, r):
        return self.k1(r) * self.k2(r)

    def _dk_dpars(self, r):
        a, b = self.k1(r), self.k2(r)
        return [d * b for d in self.k1._dk_dpars(r)] + [a * d for d in self.k2._dk_dpars(r)]

    def __repr__(self):
        return f'{self.k1} * {self.k2}'


class _unary_operator(covFunction):
    """Base of the operators on ONE kernel (covfunc.py:83-95): takes a twice-differentiable kernel, shares its parameter
    array and names.  Private in the reference too; kept so that ``isinstance(k, covfunc._unary_operator)`` holds."""

    def __init__(self, k):
        if not getattr(k, '_twice_differentiable', False):
            raise ValueError(f'kernel {k} is not twice differentiable')
        self.k = k
        self.kerneltype = 'complex_unary'
        self.pars = k.pars
        self._param_names = k._param_names
        self._tag = 'd' + k._tag


class Derivative(_unary_operator):
    """d^2 k / dx_i dx_j of a twice-differentiable kernel (covfunc.py:98-104)."""
    _derivative_ids = {}       # filled below: kernel class -> KID of its derivative

    def __call__(self, r):
        return self.k._dkdxidj(r)

    def __repr__(self):
        self.k.pars = self.pars            # the reference syncs here, and only here
        return f'd {self.k}'

    def _device_program(self):
        kid = self._derivative_ids.get(type(self.k))
        if kid is None or not _is_builtin(type(self)):
            return None
        return [(OP_PUSH, kid, 0)], np.asarray(self.k.pars, dtype=float).ravel()


# ------------------------------------------------------------------ kernels
class Constant(covFunction):
    """K_ij = c^2 (covfunc.py:107-125)."""
    _param_names = 'c',
    _tag = 'C'
    _device_id = KID['CONSTANT']

    def __init__(self, c: float):
        super().__init__(c)

    def __call__(self, r):
        return np.full_like(r, self.pars[0]**2)


class WhiteNoise(covFunction):
    """K_ij = w^2 delta_ij on a square matrix, w^2 everywhere otherwise
    (covfunc.py:128-148)."""
    _param_names = 'wn',
    _tag = 'WN'
    _device_id = KID['WHITENOISE']

    def __init__(self, w: float):
        super().__init__(w)

    def __call__(self, r):
        w2 = self.pars[0]**2
        if r.ndim == 2 and r.shape[0] == r.shape[1]:
            return w2 * np.eye(r.shape[0], dtype=r.dtype)
        return np.full_like(r, w2)


class SquaredExponential(covFunction):
    r"""K_ij = theta^2 exp(-r^2 / (2 ell^2)) (covfunc.py:151-185)."""
    _param_names = 'theta', 'ell'
    _tag = 'SE'
    _twice_differentiable = True
    _device_id = KID['SE']

    def __init__(self, theta: float, ell: float):
        super().__init__(theta, ell)

    def __call__(self, r):
        return self.pars[0]**2 * np.exp(-0.5 * r**2 / self.pars[1]**2)

    def _dk_dpars(self, r):
        theta, ell = self.pars
        K = self(r)
        return [2 * K / theta, K * r**2 / ell**3]

    def _dkdxi(self, r):
        theta, ell = self.pars
        return theta**2 * (-r) * np.exp(-0.5 * (-r)**2 / ell**2) / ell**2

    def _dkdxj(self, r):
        theta, ell = self.pars
        return theta**2 * r * np.exp(-0.5 * r**2 / ell**2) / ell**2

    def _dkdxidj(self, r):
        scale = self.pars[0]**2 / self.pars[1]**4
        poly = self.pars[1]**2 - r**2
        return scale * poly * np.exp(-0.5 * r**2 / self.pars[1]**2)


class Periodic(covFunction):
    r"""K_ij = theta^2 exp(-2 sin^2(pi |r| / P) / ell^2) (covfunc.py:188-221)."""
    _param_names = 'theta', 'P', 'ell'
    _tag = 'P'
    _twice_differentiable = True
    _device_id = KID['PERIODIC']

    def __init__(self, theta: float, P: float, ell: float):
        super().__init__(theta, P, ell)

    def __call__(self, r):
        theta, P, ell = self.pars
        return theta**2 * np.exp(-2 * np.sin(np.pi * np.abs(r) / P)**2 / ell**2)

    def _dk_dpars(self, r):
        theta, P, ell = self.pars
        K = self(r)
        x = np.pi * np.abs(r) / P
        return [2 * K / theta, K * 2 * x * np.sin(2 * x) / (P * ell**2), K * 4 * np.sin(x)**2 / ell**3]

    def _dkdxidj(self, r):
        theta, P, ell = self.pars
        x = np.pi * r / P
        scale = 4 * np.pi**2 * theta**2
        poly = ell**2 * np.cos(2 * x) - 4 * np.sin(x)**2 * np.cos(x)**2
        return scale * poly * np.exp(-2 * np.sin(x)**2 / ell**2)


class QuasiPeriodic(covFunction):
    r"""K_ij = theta^2 exp(-r^2/(2 le^2) - 2 sin^2(pi |r|/P)/lp^2)
    (covfunc.py:224-266); identical to SquaredExponential * Periodic."""
    _param_names = 'theta', 'le', 'P', 'lp'
    _tag = 'QP'
    _twice_differentiable = True
    _device_id = KID['QP']

    def __init__(self, theta: float, elle: float, P: float, ellp: float):
        super().__init__(theta, elle, P, ellp)

    def __call__(self, r):
        theta, le, P, lp = self.pars
        periodic = -2 * np.sin(np.pi * np.abs(r) / P)**2 / lp**2
        decay = r**2 / (2 * le**2)
        return theta**2 * np.exp(periodic - decay)

    def _dk_dpars(self, r):
        theta, le, P, lp = self.pars
        K = self(r)
        x = np.pi * np.abs(r) / P
        return [2 * K / theta, K * r**2 / le**3, K * 2 * x * np.sin(2 * x) / (P * lp**2),
                K * 4 * np.sin(x)**2 / lp**3]

    def _dkdxidj(self, r):
        theta, le, P, lp = self.pars
        scale = 2 * theta**2 / (P**2 * lp**4 * le**4)
        poly = P**2 * lp**4 * le**2 - \
            2 * P**2 * lp**4 * r**2 - \
            4 * np.pi * P * lp**2 * le**2 * r * np.sin(2 * np.pi * r / P) + \
            2 * np.pi**2 * lp**2 * le**4 * np.cos(2 * np.pi * r / P) - \
            8 * np.pi**2 * le**4 * np.sin(np.pi * r / P)**2 * np.cos(np.pi * r / P)**2
        envelope = np.exp(-(lp**2 * r**2 + 2 * le**2 * np.sin(np.pi * r / P)**2)
                          / (lp**2 * le**2))
        return scale * poly * envelope


Derivative._derivative_ids = {SquaredExponential: KID['DSE'],
                              Periodic: KID['DPERIODIC'],
                              QuasiPeriodic: KID['DQP']}


class RationalQuadratic(covFunction):
    """K_ij = theta^2 (1 + r^2/(2 alpha ell^2))^-alpha (covfunc.py:269-288)."""
    _param_names = 'theta', 'alpha', 'ell'
    _tag = 'RQ'
    _device_id = KID['RQ']

    def __init__(self, theta: float, alpha: float, ell: float):
        super().__init__(theta, alpha, ell)

    def __call__(self, r):
        theta, alpha, ell = self.pars
        return theta**2 * (1 + 0.5 * r**2 / (alpha * ell**2))**(-alpha)


class RQP(covFunction):
    """Periodic times rational quadratic (covfunc.py:291-313).  Positional
    order is (theta, alpha, elle, P, ellp) although the names tuple says
    otherwise -- as in the reference."""
    _param_names = 'theta', 'alpha', 'elle', 'ellp', 'P'
    _tag = 'RQP'
    _device_id = KID['RQP']

    def __init__(self, theta: float, alpha: float, elle: float, P: float,
                 ellp: float):
        super().__init__(theta, alpha, elle, P, ellp)

    def __call__(self, r):
        theta, alpha, le, P, lp = self.pars
        periodic = np.exp(-2 * np.sin(np.pi * np.abs(r) / P)**2 / lp**2)
        return theta**2 * periodic * (1 + r**2 / (2 * alpha * le**2))**(-alpha)


class Cosine(covFunction):
    """K_ij = theta^2 cos(2 pi |r| / P) (covfunc.py:316-331)."""
    _param_names = 'theta', 'P'
    _tag = 'COS'
    _device_id = KID['COSINE']

    def __init__(self, theta: float, P: float):
        super().__init__(theta, P)

    def __call__(self, r):
        return self.pars[0]**2 * np.cos(2 * np.pi * np.abs(r) / self.pars[1])


class Exponential(covFunction):
    r"""K_ij = theta^2 exp(-|r| / ell) (covfunc.py:334-352)."""
    _param_names = 'theta', 'ell'
    _tag = 'EXP'
    _device_id = KID['EXPONENTIAL']

    def __init__(self, theta: float, ell: float):
        super().__init__(theta, ell)

    def __call__(self, r):
        return self.pars[0]**2 * np.exp(-np.abs(r) / self.pars[1])


class Matern32(covFunction):
    """Matern nu = 3/2 (covfunc.py:355-373)."""
    _param_names = 'theta', 'ell'
    _tag = 'M32'
    _device_id = KID['MATERN32']

    def __init__(self, theta: float, ell: float):
        super().__init__(theta, ell)

    def __call__(self, r):
        x = np.sqrt(3.0) * np.abs(r) / self.pars[1]
        return self.pars[0]**2 * (1.0 + x) * np.exp(-x)


class Matern52(covFunction):
    """Matern nu = 5/2 (covfunc.py:376-396)."""
    _param_names = 'theta', 'ell'
    _tag = 'M52'
    _device_id = KID['MATERN52']

    def __init__(self, theta: float, ell: float):
        super().__init__(theta, ell)

    def __call__(self, r):
        a = np.abs(r)
        ell = self.pars[1]
        poly = 1.0 + (3 * np.sqrt(5) * ell * a + 5 * a**2) / (3 * ell**2)
        return self.pars[0]**2 * poly * np.exp(-np.sqrt(5.0) * a / ell)


class Linear(covFunction):
    """(t1 - c)(t2 - c) (covfunc.py:399-412).  Needs (r, t1, t2), so
    ``_KMatrix`` cannot evaluate it -- as in the reference."""

    def __init__(self, c):
        super().__init__(c)
        self.tag = 'LIN'
        self.c = c

    def __call__(self, r, t1, t2):
        return (t1 - self.pars[0]) * (t2 - self.pars[0])


class GammaExp(covFunction):
    """theta^2 exp(-(|r|/l)^gamma) (covfunc.py:415-432)."""
    _device_id = KID['GAMMAEXP']

    def __init__(self, theta, gamma, l):
        super().__init__(theta, gamma, l)
        self.tag = 'GammaExp'
        self.theta, self.gamma, self.l = theta, gamma, l

    def __call__(self, r):
        return self.pars[0]**2 * np.exp(-(np.abs(r) / self.pars[2])**self.pars[1])


class Polynomial(covFunction):
    """(a t1 t2 + b)^c (covfunc.py:435-455); called with (t1, t2)."""
    _device_id = KID['POLYNOMIAL']

    def __init__(self, theta, a, b, c):
        super().__init__(theta, a, b, c)
        self.tag = 'POLY'
        self.theta, self.a, self.b, self.c = theta, a, b, c

    def __call__(self, t1, t2):
        return (self.pars[1] * t1 * t2 + self.pars[2])**self.pars[3]

    def _device_pars(self):
        return self.pars[1:4]


class Piecewise(covFunction):
    """Third-order piecewise polynomial with compact support eta/2
    (covfunc.py:458-473)."""
    _device_id = KID['PIECEWISE']

    def __init__(self, eta):
        super().__init__(eta)
        self.eta = eta
        self.type = 'unknown'

    def __call__(self, r):
        x = np.abs(r / (0.5 * self.pars[0]))
        return np.where(x > 1, 0, (3 * x + 1) * (1 - x)**3)


class Paciorek(covFunction):
    """Stationary Paciorek kernel (covfunc.py:477-496); evaluates from the
    constructor attributes."""
    _device_id = KID['PACIOREK']

    def __init__(self, amplitude, ell_1, ell_2):
        super().__init__(amplitude, ell_1, ell_2)
        self.amplitude, self.ell_1, self.ell_2 = amplitude, ell_1, ell_2
        self.params_number = 3

    def __call__(self, r):
        s = self.ell_1**2 + self.ell_2**2
        a = np.sqrt(2 * self.ell_1 * self.ell_2 / s)
        b = np.exp(-2 * r * r / s)
        return self.amplitude**2 * a * b

    def _device_pars(self):
        return [self.amplitude, self.ell_1, self.ell_2]


class NewPeriodic(covFunction):
    """Rational quadratic mapped on the circle (covfunc.py:499-519)."""
    _device_id = KID['NEWPERIODIC']

    def __init__(self, amplitude, alpha2, P, l):
        super().__init__(amplitude, alpha2, P, l)
        self.amplitude, self.alpha2, self.P, self.l = amplitude, alpha2, P, l
        self.params_number = 4

    def __call__(self, r):
        s2 = np.sin(np.pi * np.abs(r) / self.P)**2
        a = (1 + 2 * s2 / (self.alpha2 * self.l**2))**(-self.alpha2)
        return self.amplitude**2 * a

    def _device_pars(self):
        return [self.amplitude, self.alpha2, self.P, self.l]


class QuasiNewPeriodic(covFunction):
    """NewPeriodic times squared exponential (covfunc.py:522-546)."""
    _device_id = KID['QUASINEWPERIODIC']

    def __init__(self, amplitude, alpha2, ell_e, P, ell_p):
        super().__init__(amplitude, alpha2, ell_e, P, ell_p)
        self.amplitude, self.alpha2 = amplitude, alpha2
        self.ell_e, self.P, self.ell_p = ell_e, P, ell_p
        self.params_number = 5

    def __call__(self, r):
        s2 = np.sin(np.pi * np.abs(r) / self.P)**2
        a = (1 + 2 * s2 / (self.alpha2 * self.ell_p**2))**(-self.alpha2)
        b = np.exp(-0.5 * r**2 / self.ell_e**2)
        return self.amplitude**2 * a * b

    def _device_pars(self):
        return [self.amplitude, self.alpha2, self.ell_e, self.P, self.ell_p]


class NewRQP(covFunction):
    """NewPeriodic times rational quadratic (covfunc.py:549-576).  The
    reference's ``__call__`` dies on ``np.sine`` (:574); so does this one, with
    the same exception type, instead of inventing values it never produced."""

    def __init__(self, amplitude, alpha1, alpha2, ell_e, P, ell_p):
        super().__init__(amplitude, alpha1, alpha2, ell_e, P, ell_p)
        self.amplitude, self.alpha1, self.alpha2 = amplitude, alpha1, alpha2
        self.ell_e, self.P, self.ell_p = ell_e, P, ell_p
        self.params_number = 5

    def __call__(self, r):
        raise AttributeError("module 'numpy' has no attribute 'sine'")


def _harmonic_terms(N, P, t):
    """Shared pieces of the (Quasi)HarmonicPeriodic kernels, with the
    reference's operator precedence: ``x / 2*sin(y)`` is ``(x/2)*sin(y)``."""
    phase = (N + 0.5) * 2 * np.pi * t / P
    half = np.pi * t / P
    s_term = np.sin(phase) / 2 * np.sin(half)
    c_term = np.cos(phase) / 2 * np.sin(half)
    cot = 0.5 / np.tan(half)
    return s_term, cot - c_term


class HarmonicPeriodic(covFunction):
    """Periodic kernel with N harmonics (covfunc.py:579-607); called with
    (t1, t2) and left without nugget by ``_KMatrix``."""
    _device_id = KID['HARMONICPERIODIC']

    def __init__(self, N, amplitude, P, ell):
        super().__init__(N, amplitude, P, ell)
        self.N, self.amplitude, self.ell, self.P = N, amplitude, ell, P
        self.params_number = 4

    def __call__(self, t1, t2):
        s1, u1 = _harmonic_terms(self.N, self.P, t1)
        s2, u2 = _harmonic_terms(self.N, self.P, t2)
        dist2 = (s1 - s2)**2 + (u1 - u2)**2
        return self.amplitude**2 * np.exp(-0.5 * dist2 / self.ell**2)

    def _device_pars(self):
        return [self.N, self.amplitude, self.P, self.ell]


class QuasiHarmonicPeriodic(covFunction):
    """HarmonicPeriodic times squared exponential (covfunc.py:610-642);
    ``pars`` holds (amplitude, ell_e, P, ell_p) -- N is an attribute only."""
    _device_id = KID['QUASIHARMONICPERIODIC']

    def __init__(self, N, amplitude, ell_e, P, ell_p):
        super().__init__(amplitude, ell_e, P, ell_p)
        self.N, self.amplitude = N, amplitude
        self.ell_e, self.P, self.ell_p = ell_e, P, ell_p
        self.params_number = 5

    def __call__(self, t1, t2):
        s1, u1 = _harmonic_terms(self.N, self.P, t1)
        s2, u2 = _harmonic_terms(self.N, self.P, t2)
        dist2 = (s1 - s2)**2 + (u1 - u2)**2
        a = np.exp(-0.5 * dist2 / self.ell_p**2)
        b = np.exp(-0.5 * (t1 - t2)**2 / self.ell_e**2)
        return self.amplitude**2 * a * b

    def _device_pars(self):
        return [self.N, self.amplitude, self.ell_e, self.P, self.ell_p]


class CosPeriodic(covFunction):
    """exp(-2 cos^2(pi |r| / P) / ell^2) (covfunc.py:645-665).  ``pars`` holds
    only (P, ell): the amplitude never enters the parameter vector."""
    _device_id = KID['COSPERIODIC']

    def __init__(self, amplitude, P, ell):
        super().__init__(P, ell)
        self.amplitude, self.ell, self.P = amplitude, ell, P
        self.params_number = 3

    def __call__(self, r):
        c2 = np.cos(np.pi * np.abs(r) / self.P)**2
        return self.amplitude**2 * np.exp(-2 * c2 / self.ell**2)

    def _device_pars(self):
        return [self.amplitude, self.P, self.ell]


class QuasiCosPeriodic(covFunction):
    """CosPeriodic times squared exponential (covfunc.py:668-689)."""
    _device_id = KID['QUASICOSPERIODIC']

    def __init__(self, amplitude, ell_e, P, ell_p):
        super().__init__(amplitude, ell_e, P, ell_p)
        self.amplitude, self.ell_e, self.P, self.ell_p = amplitude, ell_e, P, ell_p
        self.params_number = 4

    def __call__(self, r):
        c2 = np.cos(np.pi * np.abs(r) / self.P)**2
        return self.amplitude**2 * np.exp(-2 * c2 / self.ell_p**2
                                          - r**2 / (2 * self.ell_e**2))

    def _device_pars(self):
        return [self.amplitude, self.ell_e, self.P, self.ell_p]
