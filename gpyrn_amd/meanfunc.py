"""Mean functions of the GPRN outputs.

Plugin surface of the reference's ``gpyrn/meanfunc.py`` (base :9-46,
``Sum``/``Product`` :49-117, concrete means :120-273).  Mean functions are
O(p N) host work evaluated once per ``ELBOcalc`` (meanfield.py:382-411,623);
they stay NumPy on purpose -- their result, the
mean-subtracted data, is what gets uploaded (``gprn_set_y_resid``).
"""
import numpy as np

from ._utils import Array, _array_input, _take_leading  # noqa: F401

__all__ = [
    'Constant', 'MultiConstant', 'Linear', 'Parabola', 'Cubic', 'Sine',
]


class meanFunction:
    """Base class: ``pars`` vector, chained ``set_parameters``, ``+`` / ``*``."""
    _parsize = 0

    def __init__(self, *pars):
        self.pars = np.array(pars, dtype=float)

    def __repr__(self):
        inner = ', '.join(str(v) for v in self.pars)
        return f'{type(self).__name__}({inner})'

    def get_parameters(self):
        return self.pars

    @_array_input
    def set_parameters(self, p):
        return _take_leading(self, p, 'mean')

    def __add__(self, other):
        return Sum(self, other)

    __radd__ = __add__

    def __mul__(self, other):
        return Product(self, other)

    __rmul__ = __mul__


class _pair(meanFunction):
    """Two means combined; parameters are the concatenation of both."""

    def __init__(self, m1, m2):
        self.m1, self.m2 = m1, m2
        self._param_names = tuple(self._names(m1, m2))
        self._parsize = m1._parsize + m2._parsize
        self.pars = np.r_[m1.pars, m2.pars]

    @staticmethod
    def _names(m1, m2):
        return list(m1._param_names) + list(m2._param_names)

    @_array_input
    def set_parameters(self, p):
        n = self.pars.size
        assert len(p) >= n, f'too few parameters for mean {type(self).__name__}'
        self.pars = np.array(p[:n], dtype=float) if len(p) > n else p
        rest = self.m1.set_parameters(p)
        rest = self.m2.set_parameters(rest)
        return rest if len(p) > n else None


class Sum(_pair):
    """m1 + m2; equal classes get numbered parameter names (meanfunc.py:49-82)."""

    @staticmethod
    def _names(m1, m2):
        if m1.__class__ == m2.__class__:
            return [f'{n}1' for n in m1._param_names] + \
                   [f'{n}2' for n in m2._param_names]
        return list(m1._param_names) + list(m2._param_names)

    def __repr__(self):
        return f'{self.m1} + {self.m2}'

    @_array_input
    def __call__(self, t):
        return self.m1(t) + self.m2(t)


class Product(_pair):
    """m1 * m2 (meanfunc.py:85-117)."""

    def __repr__(self):
        return f'{self.m1} * {self.m2}'

    @_array_input
    def __call__(self, t):
        return self.m1(t) * self.m2(t)


class Constant(meanFunction):
    """m(t) = c (meanfunc.py:120-135)."""
    _param_names = 'c',
    _parsize = 1

    def __init__(self, c: float):
        super().__init__(c)

    @_array_input
    def __call__(self, t):
        return np.full(t.shape, self.pars[0])


class MultiConstant(meanFunction):
    """Per-instrument offsets relative to the last instrument, whose average
    is the final parameter (meanfunc.py:138-187).

    Args:
        offsets: [off_1, ..., off_{n-1}, mean_n]
        obsid: one-based instrument index of every observation
        time: observation times, same size as ``obsid``
    """
    _parsize = 0

    def __init__(self, offsets: np.ndarray, obsid: np.ndarray, time: np.ndarray):
        self.obsid, self.time = obsid, time
        self._parsize = (np.ediff1d(obsid) == 1).sum() + 1
        self.ii = obsid.astype(int) - 1
        if isinstance(offsets, float):
            offsets = [offsets]
        assert len(offsets) == self._parsize, \
            f'wrong number of parameters, expected {self._parsize} got {len(offsets)}'
        super().__init__(*offsets)
        self._param_names = [f'off{i}' for i in range(1, self._parsize)] + ['mean']

    def time_bins(self):
        """Edges between instruments: midpoints of the gaps, plus the start."""
        last_of_block = self.time[np.ediff1d(self.obsid, 0, None) != 0]
        first_of_next = self.time[np.ediff1d(self.obsid, None, 0) != 0]
        mid = np.mean((last_of_block, first_of_next), axis=0)
        return np.sort(np.r_[self.time[0], mid])

    @_array_input
    def __call__(self, t):
        offsets = np.pad(self.pars[:-1], (0, 1))
        if t.size == self.time.size:
            which = self.ii
        else:
            which = np.digitize(t, self.time_bins()) - 1
        return np.full_like(t, self.pars[-1]) + np.take(offsets, which)


class Linear(meanFunction):
    """m(t) = slope (t - mean(t)) + intercept (meanfunc.py:190-208)."""
    _param_names = ('slope', 'intercept')
    _parsize = 2

    def __init__(self, slope: float, intercept: float):
        super().__init__(slope, intercept)

    @_array_input
    def __call__(self, t):
        return self.pars[0] * (t - t.mean()) + self.pars[1]


class Parabola(meanFunction):
    """m(t) = quad t^2 + slope t + intercept (meanfunc.py:211-229; the names
    tuple is ordered as in the reference)."""
    _param_names = ('slope', 'intercept', 'quadratic')
    _parsize = 3

    def __init__(self, quad: float, slope: float, intercept: float):
        super().__init__(quad, slope, intercept)

    @_array_input
    def __call__(self, t):
        return np.polyval(self.pars, t)


class Cubic(meanFunction):
    """m(t) = cub t^3 + quad t^2 + slope t + intercept (meanfunc.py:232-251)."""
    _param_names = ('cub', 'quad', 'slope', 'intercept')
    _parsize = 4

    def __init__(self, cub: float, quad: float, slope: float, intercept: float):
        super().__init__(cub, quad, slope, intercept)

    @_array_input
    def __call__(self, t):
        return np.polyval(self.pars, t)


class Sine(meanFunction):
    """m(t) = amplitude sin(2 pi t / period + phase) (meanfunc.py:254-273)."""
    _param_names = ('amplitude', 'period', 'phase')
    _parsize = 3

    def __init__(self, amplitude: float, period: float, phase: float):
        super().__init__(amplitude, period, phase)

    @_array_input
    def __call__(self, t):
        amplitude, period, phase = self.pars
        return amplitude * np.sin((2 * np.pi * t / period) + phase)
