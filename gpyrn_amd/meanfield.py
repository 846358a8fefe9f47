"""Mean-field variational inference for GP regression networks on MI355X.

Drop-in for the reference's ``gpyrn.meanfield.inference`` on its hot path:
``ELBO`` / ``ELBOcalc`` / ``ELBOaux`` / ``nELBO`` / ``optimize`` and the whole
component / parameter API (meanfield.py:92-379, 556-710, 1095-1152), same
names, argument meaning, return shapes and error types.  What differs is where
the work happens: covariance assembly, every Cholesky / triangular inverse and
all ELBO reductions run in libgprn_hip.so (include/gprn_hip.h) on the GPU,
device-resident across sweeps; per sweep only the ELBO scalar returns to the
host for the reference's stop rule (meanfield.py:640-646).

There is no CPU fallback: without the built library and a GPU, anything that
needs the ELBO raises ``_hip.BackendUnavailable``.  Host-side work that remains
is O(#parameters) bookkeeping, O(pN) mean functions and the O(pqN)
``_initMuVar`` start point.

Reference behaviours kept because results must match: the
first ``ELBOaux`` of ``ELBOcalc`` is evaluated and discarded (its ELBO is
``elboArray[0]``); ``max_iter`` defaults to 10000; ``_mu``/``_var`` are cached only
on the converged return; ``_initMuVar`` uses the first p weight amplitudes and
emits the weight block node-major; ELBO = (LogL + LogP + Ent) / q.
"""
import os
import time as time_module
from itertools import chain

import numpy as np

from . import _hip, covfunc, meanfunc, sharding
from ._utils import Array, _array_input  # noqa: F401

__all__ = ['inference']

_NUGGET = 1e-6          # meanfield.py:433
_TINY_NUGGET = 1.25e-12 # meanfield.py:450
_TWO_ARGUMENT = (covfunc.HarmonicPeriodic, covfunc.QuasiHarmonicPeriodic,
                 covfunc.Polynomial)          # meanfield.py:426-428
_NOT_SET = 'GPRN components not set, use set_components'


class inference:
    """
    Mean-field variational inference for GPRNs (Nguyen & Bonilla 2013).

    Args:
        q: int
            Number of latent node functions f(x)
        time: array
            Time coordinates
        *args: arrays
            The observed data in the following order:
                y1, y1error, y2, y2error, ...
        device: int, keyword only
            GPU ordinal (default: LOCAL_RANK when sharded, else 0)
        comm: ``sharding.Comm``, keyword only
            Shard the q + q*p latent GPs over the ranks of one node
    """

    def __init__(self, q: int, time: Array, *args, device=None, comm=None):
        self.q = q
        self.time = time
        self.N = self.time.size

        assert len(args) > 0 and len(args) % 2 == 0, \
            'Number of observed data arrays should be even: y1, y1error, ...'
        assert np.all(np.array([len(a) for a in args]) == self.N), \
            'Output arrays should all have the same dimensions as time'

        self.p = int(len(args) / 2)
        self.qp = self.q * self.p
        self.d = self.N * self.q * (self.p + 1)

        self.tt = np.tile(time, self.p)                 # "extended" time
        self.y = np.concatenate([args[::2]])            # (p, N)
        self.yerr = np.concatenate([args[1::2]])
        self.yerr2 = self.yerr**2

        self._components_set = False
        self._frozen_mask = np.array([])
        self._mu, self._var = None, None
        self._mu_var_iters = 0
        self.update_muvar_after = 50
        self.elbo_max_iter = 5000

        self._device = device
        self._comm = comm
        self._ctx = None
        self._prior_key = None
        self.last_info = 0

    # ------------------------------------------------------------ components
    def set_components(self, nodes, weights, means, jitters):
        """
        Set the GPRN components: q node kernels, q*p weight kernels (flat
        order node-major: weight of node j and output i at index j*p + i),
        p mean functions and p jitters.  Bare objects are accepted for
        single-element lists.
        """
        if isinstance(nodes, covfunc.covFunction):
            nodes = [nodes]
        if len(nodes) != self.q:
            raise ValueError('Wrong number of nodes provided, '
                             f'expected {self.q} got {len(nodes)}')
        if isinstance(weights, covfunc.covFunction):
            weights = [weights]
        if len(weights) != self.qp:
            raise ValueError('Wrong number of weights provided, '
                             f'expected {self.qp} got {len(weights)}')
        if isinstance(means, (int, float, meanfunc.meanFunction)):
            means = [means]
        if isinstance(jitters, (int, float)):
            jitters = [jitters]

        self.nodes = nodes
        self.weights = weights
        self.means = means
        self.jitters = np.array(jitters, dtype=float)
        self._components_set = True

    def _component_chain(self):
        return chain.from_iterable([self.nodes, self.weights, self.means])

    def get_parameters(self, nodes=None, weights=None, means=None,
                       jitters=None, include_frozen=False):
        """ Values of all the GPRN parameters: nodes, weights, means, jitters """
        given = (nodes, weights, means, jitters)
        if not self._components_set and all(g is None for g in given):
            raise ValueError('Cannot get parameters. '
                             'Provide arguments or run set_components before.')
        if self._components_set:
            nodes, weights, means, jitters = (self.nodes, self.weights,
                                              self.means, self.jitters)
        pieces = []
        for group in (nodes, weights, means):
            if group is not None:
                pieces += [c.get_parameters() for c in group]
        if jitters is not None:
            pieces += [np.array([j]) for j in jitters]
        flat = np.concatenate(pieces).ravel()
        return flat if include_frozen else flat[~self.frozen_mask]

    @_array_input
    def set_parameters(self, parameters: Array):
        """ Set all parameters (full vector, or only the non-frozen ones) """
        assert self._components_set, _NOT_SET
        current = self.get_parameters(include_frozen=True)
        n_all = self.n_parameters
        n_free = n_all - self.frozen_mask.sum()

        if parameters.size == n_all:
            parameters[self.frozen_mask] = current[self.frozen_mask]
        elif parameters.size == n_free:
            for i, value in enumerate(current):
                if self.frozen_mask[i]:
                    parameters = np.insert(parameters, i, value)
        else:
            msg = f'Wrong number of parameters provided: got {parameters.size}, '
            msg += f'expected {n_all}' if n_all == n_free else \
                f'expected {n_all} (all) or {n_free} (not frozen)'
            raise ValueError(msg)

        for component in self._component_chain():
            parameters = component.set_parameters(parameters)
        self.jitters = parameters

    @property
    def n_parameters(self):
        """ Total number of parameters """
        assert self._components_set, _NOT_SET
        return sum(c.pars.size for c in self._component_chain()) + self.jitters.size

    @property
    def parameters_dict(self):
        """ Dictionary with parameters names and values """
        assert self._components_set, _NOT_SET
        out = {}
        for label, group in (('node', self.nodes), ('weight', self.weights),
                             ('mean', self.means)):
            for i, comp in enumerate(group, start=1):
                for name, value in zip(comp._param_names, comp.pars):
                    out[f'{label}{i}.{name}'] = value
        for i, jit in enumerate(self.jitters, start=1):
            out[f'jitter{i}'] = jit
        return out

    def _set_frozen(self, value, index, name):
        self.frozen_mask        # materialise the mask
        if index is None and name is None:
            raise ValueError('Provide either index or name')
        if name is None:
            self._frozen_mask[index] = value
        elif index is None:
            names = list(self.parameters_dict.keys())
            if '*' in name:
                stem = name.replace('*', '')
                for i, known in enumerate(names):
                    if stem in known:
                        self._frozen_mask[i] = value
            else:
                assert name in names, f'Name "{name}" not found in parameters_dict'
                self._frozen_mask[names.index(name)] = value

    def freeze_parameter(self, index=None, name=None):
        """ Freeze (do not fit for) a parameter by index or name; a "*" in
        `name` matches every parameter whose name contains the rest. """
        self._set_frozen(True, index, name)

    def thaw_parameter(self, index=None, name=None):
        """ Thaw (free) a parameter by index or name ("*" as in freeze). """
        self._set_frozen(False, index, name)

    def freeze_all_parameters(self):
        self._frozen_mask = np.ones(self._frozen_mask.size, dtype=bool)

    def thaw_all_parameters(self):
        self._frozen_mask = np.zeros(self._frozen_mask.size, dtype=bool)

    fix_parameter = freeze_parameter
    fix_all_parameters = freeze_all_parameters
    free_parameter = thaw_parameter
    free_all_parameters = thaw_all_parameters

    @property
    def frozen_mask(self):
        """ Boolean mask for the frozen parameters """
        assert self._components_set, _NOT_SET
        if self._frozen_mask.size == 0:
            self._frozen_mask = np.full(self.n_parameters, False, dtype=bool)
        return self._frozen_mask

    @frozen_mask.setter
    def frozen_mask(self, mask):
        raise NotImplementedError(
            'Do not set frozen_mask, use thaw_parameter/freeze_parameter')

    # ------------------------------------------------------------- host glue
    def _mean(self, means, time=None):
        """ Mean functions evaluated at `time` (default: the data), flat (p*N,) """
        t = self.time if time is None else time
        n = t.size
        m = np.zeros(n * self.p)
        for i, fun in enumerate(means):
            if fun is not None:
                m[i * n:(i + 1) * n] = fun(t)
        return m

    def _u_to_fhatW(self, u):
        """ Split a flat variational vector: nodes (1,q,N), weights (p,q,N) """
        f = u[:self.q * self.N].reshape((1, self.q, self.N))
        w = u[self.q * self.N:].reshape((self.p, self.q, self.N))
        return f, w

    def _initMuVar(self, nodes, weights, jitter):
        """ Data-driven start point of the coordinate ascent (meanfield.py:491-510) """
        jitter = np.asarray(jitter, dtype=float)
        w_amp = np.array([w.pars[0] for w in weights][:self.p])[:, None]
        absy, sgn = np.abs(self.y), np.sign(self.y)
        mu_f, mu_w, var_f, var_w = [], [], [], []
        for node in nodes:
            a = node.pars[0]
            mu_f.append(np.mean(np.sqrt(absy * a / w_amp) * sgn, axis=0))
            mu_w.append(np.sqrt(absy * w_amp / a))
            var_f.append(np.full(self.N, np.mean(jitter)))
            var_w.append(jitter[:, None] * np.ones((self.p, self.N)))
        mu = np.concatenate((mu_f, mu_w), axis=None)
        var = np.concatenate((var_f, var_w), axis=None)
        return mu, var

    def _randomMuVar(self):
        return np.random.randn(self.d, 1), np.random.rand(self.d, 1)

    def _get_components(self, nodes=None, weights=None, means=None,
                        jitters=None):
        if all(i is None for i in (nodes, weights, means, jitters)) \
                and not self._components_set:
            raise ValueError(_NOT_SET)
        nodes = self.nodes if nodes is None else nodes
        weights = self.weights if weights is None else weights
        means = self.means if means is None else means
        jitters = self.jitters if jitters is None else jitters
        return nodes, weights, means, jitters

    # ---------------------------------------------------------------- device
    def _backend(self):
        """The GPU context of this problem (created on first use, after fork)."""
        if self._ctx is None:
            comm = self._comm
            device = self._device
            if device is None:
                # one rank per GPU; with fewer GPUs than ranks (rehearsals) ranks share devices
                device = comm.local_rank % max(1, _hip.device_count()) if comm is not None else 0
            ctx = _hip.Context(device)
            forced = bool(os.environ.get('GPRN_FORCE_RCCL'))      # one-rank communicator, for tests
            if comm is not None and (comm.world > 1 or forced):
                ctx.comm_init(comm.world, comm.rank, comm.unique_id())
                comm.done()
            ctx.set_data(np.asarray(self.time, dtype=float), self.y, self.yerr, self.q)
            if comm is not None and comm.world > 1:
                ctx.set_owners(sharding.owners(self.p, self.q, comm.world))
            self._ctx = ctx
        return self._ctx

    def _host_K(self, kernel, time):
        """Kernel matrix evaluated in Python: the path for user-defined kernels."""
        if isinstance(kernel, _TWO_ARGUMENT):
            return kernel(time[:, None], time[None, :])
        r = time[:, None] - time[None, :]
        return kernel(r) + _NUGGET * np.eye(time.size)

    def _kernel_spec(self, kernel):
        """How a kernel reaches the device: a postfix program for the fused fill
        (built-ins and their sums/products), or a host-evaluated matrix."""
        program = kernel._device_program() if isinstance(kernel, covfunc.covFunction) else None
        if program is None:
            K = np.asarray(self._host_K(kernel, np.asarray(self.time, dtype=float)), dtype=float)
            return ('host', K)
        ops, params = program
        return ('device', tuple(ops), np.asarray(params, dtype=float),
                not isinstance(kernel, _TWO_ARGUMENT))

    @staticmethod
    def _spec_key(spec):
        if spec[0] == 'host':
            return ('host', spec[1].tobytes())
        return ('device', spec[1], spec[2].tobytes(), spec[3])

    @staticmethod
    def _send_spec(ctx, gp, spec):
        if spec[0] == 'host':
            ctx.upload_K(gp, spec[1])
        else:
            ctx.set_kernel(gp, spec[1], spec[2], spec[3])

    def _device_K(self, kernel, time, nugget):
        """K = kernel at the data times + nugget on the diagonal through the fused HIP fill (built-in kernels
        and their sums / products at the data's own time stamps), else None."""
        if not (time is self.time or (np.shape(time) == np.shape(self.time)
                                      and np.array_equal(time, self.time))):
            return None
        program = kernel._device_program() if isinstance(kernel, covfunc.covFunction) else None
        if program is None:
            return None
        ops, params = program
        two = isinstance(kernel, _TWO_ARGUMENT)        # evaluated without a nugget (meanfield.py:426-431)
        return self._backend().eval_kernel(ops, params, 0.0 if two else nugget)

    def _KMatrix(self, kernel, time=None):
        """
        Covariance matrix of `kernel` at the data times, with the reference's
        1e-6 nugget (meanfield.py:413-434).  Built by the fused HIP fill kernel
        for built-in kernels when `time` is the data's own time vector.
        """
        time = self.time if time is None else time
        K = self._device_K(kernel, time, _NUGGET)
        return self._host_K(kernel, np.asarray(time, dtype=float)) if K is None else K

    def _tinyNuggetKMatrix(self, kernel, time=None):
        """Covariance matrix with the tiniest stability nugget, 1.25e-12 (meanfield.py:436-452)."""
        time = self.time if time is None else time
        K = self._device_K(kernel, time, _TINY_NUGGET)
        if K is not None:
            return K
        time = np.asarray(time, dtype=float)
        if isinstance(kernel, _TWO_ARGUMENT):
            return kernel(time[:, None], time[None, :])
        r = time[:, None] - time[None, :]
        return kernel(r) + _TINY_NUGGET * np.eye(time.size)

    def _predictKMatrix(self, kernel, time):
        """Cross-covariance between new times and the data times (meanfield.py:455-471; _gp.py:52-63 for
        the two-argument kernels), evaluated in Python: the path of user-defined kernels in prediction."""
        time = np.atleast_1d(np.asarray(time, dtype=float))
        data_t = np.asarray(self.time, dtype=float)
        if isinstance(kernel, _TWO_ARGUMENT):
            return kernel(time[:, None], data_t[None, :])
        return kernel(time[:, None] - data_t[None, :])

    def _sample_from_gp(self, kernel, time=None):
        """
        One draw from the GP prior of `kernel` at `time` (meanfield.py:517-531).  The reference hands the
        tiny-nugget matrix to scipy's ``multivariate_normal(..., allow_singular=True)``; here the draw is
        ``L z`` with ``z`` from NumPy's global generator and ``K + nugget I = L L^T`` factored on the GPU,
        the nugget growing from 1.25e-12 by factors of 100 (to 1.25e-6 at most) while fp64 finds the matrix
        not positive definite -- what ``allow_singular`` papers over on the CPU.  Same distribution, not
        the same stream of numbers.  Kernels without a device program are factored on the host.
        """
        time = self.time if time is None else time
        n = np.size(time)
        z = np.random.standard_normal(n)
        program = kernel._device_program() if isinstance(kernel, covfunc.covFunction) else None
        same_t = time is self.time or (np.shape(time) == np.shape(self.time) and np.array_equal(time, self.time))
        nugget = _TINY_NUGGET
        while True:
            if program is not None and same_t and not isinstance(kernel, _TWO_ARGUMENT):
                out, info = self._backend().sample_prior(program[0], program[1], nugget, z[None])
                if info == 0:
                    return out[0]
            else:
                K = self._tinyNuggetKMatrix(kernel, np.asarray(time, dtype=float))
                K = K + (nugget - _TINY_NUGGET) * np.eye(n)
                try:
                    return np.linalg.cholesky(K) @ z
                except np.linalg.LinAlgError:
                    pass
            if nugget >= 1e-6:
                raise np.linalg.LinAlgError('prior covariance is not positive definite even with a 1e-6 nugget')
            nugget *= 100.0

    def sample(self, time=None):
        """Draws of every node and weight GP from its prior (meanfield.py:533-539; it prints the two
        shapes, so does this).  Returns (node_samples (q, N), weight_samples (q*p, N))."""
        nodes, weights, means, jitters = self._get_components()
        node_samples = np.array([self._sample_from_gp(node) for node in nodes])
        weight_samples = np.array([self._sample_from_gp(weight) for weight in weights])
        print(node_samples.shape)
        print(weight_samples.shape)
        return node_samples, weight_samples

    def _setup_device(self, nodes, weights, means, jitters):
        """The setup block of ELBOcalc (meanfield.py:618-624) on the GPU."""
        ctx = self._backend()
        specs = [self._kernel_spec(k) for k in chain(nodes, weights)]
        key = tuple(self._spec_key(s) for s in specs)
        if key != self._prior_key:             # unchanged hyper-parameters keep their factors
            self._prior_key = None             # (nothing names the device's kernels until the set-up has succeeded)
            for gp, spec in enumerate(specs):
                self._send_spec(ctx, gp, spec)
            self.last_info = ctx.factor_priors()
            self._prior_key = key
        y = np.concatenate(self.y) - self._mean(means)
        ctx.set_y_resid(np.array(np.array_split(y, self.p)))
        ctx.set_jitters(np.asarray(jitters, dtype=float))
        return ctx

    # ------------------------------------------------------------------ ELBO
    @property
    def ELBO(self):
        """ The evidence lower bound for the GPRN """
        return self.ELBOcalc()[0]

    def ELBOcalc(self, nodes=None, weights=None, means=None, jitters=None,
                 max_iter=None, mu=None, var=None):
        """
        Calculate the evidence lower bound by coordinate ascent on the
        variational means/variances.

        Args:
            nodes, weights, means, jitters: optional overrides of the components
            max_iter: int, default 10000
            mu, var: arrays (flat d or (p+1,q,N)), or 'init', 'random', 'previous'

        Returns:
            ELBO (float), mu (p+1,q,N), var (p+1,q,N), iterNumber (int)
        """
        nodes, weights, means, jitters = self._get_components(
            nodes, weights, means, jitters)

        if mu is None or var is None:
            mu = var = 'init'
        mu_s = mu if isinstance(mu, str) else None
        var_s = var if isinstance(var, str) else None
        if mu_s == 'previous' or var_s == 'previous':
            if self._mu is not None:
                mu, var = self._mu, self._var
            else:
                mu, var = self._initMuVar(nodes, weights, jitters)
        elif mu_s == 'random' and var_s == 'random':
            mu, var = self._randomMuVar()
        elif mu_s == 'init' and var_s == 'init':
            mu, var = self._initMuVar(nodes, weights, jitters)

        if max_iter is None:
            max_iter = 10000

        # meanfield.py:618-649 in ONE call of the library: the set-up (when a hyper-parameter of a kernel has changed:
        # unchanged ones keep their factors), y - mean and the jitters, the starting state, then the loop -- the first
        # sweep's update thrown away and only its ELBO kept, sweeps to the stop rule or max_iter
        ctx = self._backend()
        specs = [self._kernel_spec(k) for k in chain(nodes, weights)]
        key = tuple(self._spec_key(s) for s in specs)
        setup = key != self._prior_key
        if setup:
            # (from here to the end of a successful set-up the device holds kernels -- possibly factors -- that no key
            # names: a call that raises in between must not leave the old key standing for them)
            self._prior_key = None
            for gp, spec in enumerate(specs):
                self._send_spec(ctx, gp, spec)
            self.last_info = 0
        y = np.concatenate(self.y) - self._mean(means)
        history, iterNumber, converged, info, mu, var = ctx.elbocalc(
            max_iter, setup=setup, y_resid=y, jitters=np.asarray(jitters, dtype=float),
            mu=np.asarray(mu, dtype=float), var=np.asarray(var, dtype=float))
        if setup:
            self._prior_key = key
        self.last_info = self.last_info or info
        self._elbo_history = history
        ELBO = np.float64(history[-1])
        if converged:
            self._mu, self._var = mu, var
            return ELBO, mu, var, iterNumber

        print('\nMax iterations reached')
        return ELBO, mu, var, iterNumber

    def ELBOaux(self, Kf, Kw, Lf, Lw, y, jitt2, mu, var):
        """
        One coordinate-ascent sweep + ELBO from explicit host matrices
        (compatibility with meanfield.py:651-710; `Lf`/`Lw` are recomputed on the
        device and ignored).  Returns ELBO, new_mu, new_var, sigmaF (q,N,N),
        sigmaW (q,p,N,N).
        """
        ctx = self._backend()
        Kf = np.asarray(Kf, dtype=float).reshape(self.q, self.N, self.N)
        Kw = np.asarray(Kw, dtype=float).reshape(self.qp, self.N, self.N)
        for gp, K in enumerate(chain(Kf, Kw)):
            ctx.upload_K(gp, K)
        self._prior_key = None
        self.last_info = ctx.factor_priors()
        ctx.set_y_resid(np.asarray(y, dtype=float).reshape(self.p, self.N))
        ctx.set_jitters(np.sqrt(np.asarray(jitt2, dtype=float)))
        ctx.set_muvar(np.asarray(mu, dtype=float), np.asarray(var, dtype=float))
        ctx.keep_sigma(True)
        try:
            e, _, info = ctx.sweep(1, commit=True)
            sigmaF = np.array([ctx.get_matrix(_hip.M_SIGMA, j) for j in range(self.q)])
            sigmaW = np.array([ctx.get_matrix(_hip.M_SIGMA, self.q + k)
                               for k in range(self.qp)]).reshape(self.q, self.p, self.N, self.N)
        finally:
            ctx.keep_sigma(False)
        self.last_info = self.last_info or info
        new_mu, new_var = ctx.get_muvar()
        return np.float64(e[0]), new_mu, new_var, sigmaF, sigmaW

    # ------------------------------------------------- the four step methods ELBOaux is made of (meanfield.py:713, 895, 992, 1069)
    # Private in the reference, but part of its class: a script that calls them finds them here with the reference's
    # signatures and return shapes, computed on the device -- the sweep itself for _updateSigMu, the factorisation for
    # _entropy, the resident factors plus one triangular product per latent GP for _expectedLogPrior, the ELBO assembly's own
    # kernel for _expectedLogLike.  Each invalidates the object's cached set-up (the next ELBOcalc refactors).
    def _updateSigMu(self, Kf, Kw, Lf, Lw, y, jitt2, muF, varF, muW, varW):
        """Closed-form updates of the variational covariances and means (meanfield.py:713-893; eqs. 16-19 of Nguyen &
        Bonilla 2013): ``(sigma_f (q, N, N), mu_f (q, N), sigma_w (q, p, N, N), mu_w (p, q, N))`` from the prior
        matrices, ``y - mean``, the squared jitters and the current state split as ``_u_to_fhatW`` splits it.  One device
        sweep with the explicit covariances kept (as ``ELBOaux``); ``Lf`` / ``Lw`` are recomputed there."""
        q, p, N = self.q, self.p, self.N
        mu = np.concatenate((np.reshape(muF, (1, q, N)), np.reshape(muW, (p, q, N))))
        var = np.concatenate((np.reshape(varF, (1, q, N)), np.reshape(varW, (p, q, N))))
        _, new_mu, _, sigma_f, sigma_w = self.ELBOaux(Kf, Kw, Lf, Lw, y, jitt2, mu, var)
        return sigma_f, new_mu[0], sigma_w, new_mu[1:]

    def _entropy(self, sigma_f, sigma_w):
        """Entropy of the variational distribution (meanfield.py:1069-1093): ``sum log diag chol(Sigma)`` over the q + q p
        covariances ``+ q (p + 1) N (1 + log 2 pi) / 2``.  The Choleskys run on the device (the covariances go in as the
        latent GPs' matrices, the set-up's factorisation returns ``log det``); a covariance that is not positive definite
        gives NaN, as jax's cholesky does."""
        q, p, N = self.q, self.p, self.N
        sigma_f = np.asarray(sigma_f, dtype=float).reshape(q, N, N)
        sigma_w = np.asarray(sigma_w, dtype=float).reshape(q, p, N, N)
        ctx = self._backend()
        self._prior_key = None
        for j in range(q):
            ctx.upload_K(j, sigma_f[j])
            for i in range(p):
                ctx.upload_K(q + j * p + i, sigma_w[j, i])
        info = ctx.factor_priors()
        self.last_info = info
        if info:
            return np.float64(np.nan)
        return np.float64(0.5 * np.sum(ctx.get_logdet_K()) + 0.5 * q * (p + 1) * N * (1.0 + np.log(2.0 * np.pi)))

    def _expectedLogPrior(self, Kf, Kw, Lf, Lw, sigma_f, mu_f, sigma_w, mu_w):
        """Expectation of the log prior under q(f, w) (meanfield.py:992-1067; eq. 15): per latent GP
        ``-log det K / 2 - (m^T K^-1 m + tr(K^-1 S)) / 2``, node j with the CUMULATIVE ``S = Sigma_f0 + ... + Sigma_fj``
        (:1025, 1039: quirk Q1), weight (j, i) with the raw reshape ``mu_w.reshape(q, p, N)[j, i]`` (:1021: quirk Q2),
        ``- N q (p + 1) log(2 pi) / 2``.  The matrices are factored on the device (``Lf`` / ``Lw`` are recomputed there) and
        every term comes from the resident factor (``gprn_prior_terms``: tr(K^-1 S) as one triangular product, N^3)."""
        q, p, N = self.q, self.p, self.N
        Kf = np.asarray(Kf, dtype=float).reshape(q, N, N)
        Kw = np.asarray(Kw, dtype=float).reshape(q * p, N, N)
        sigma_f = np.asarray(sigma_f, dtype=float).reshape(q, N, N)
        sigma_w = np.asarray(sigma_w, dtype=float).reshape(q, p, N, N)
        mf = np.asarray(mu_f, dtype=float).reshape(q, N)
        mw = np.asarray(mu_w, dtype=float).reshape(q, p, N)              # quirk Q2
        ctx = self._backend()
        self._prior_key = None
        for gp, K in enumerate(chain(Kf, Kw)):
            ctx.upload_K(gp, K)
        info = ctx.factor_priors()
        self.last_info = info
        if info:
            return np.float64(np.nan)
        total = 0.0
        cumulative = np.zeros((N, N))
        for j in range(q):
            cumulative = cumulative + sigma_f[j]                          # quirk Q1
            ld, quad, tr = ctx.prior_terms(j, cumulative, mf[j])
            total += -0.5 * ld - 0.5 * (quad + tr)
            for i in range(p):
                ld, quad, tr = ctx.prior_terms(q + j * p + i, sigma_w[j, i], mw[j, i])
                total += -0.5 * ld - 0.5 * (quad + tr)
        return np.float64(total - 0.5 * N * q * (p + 1) * np.log(2.0 * np.pi))

    def _expectedLogLike(self, y, jitt2, sigma_f, mu_f, sigma_w, mu_w):
        """Expected log-likelihood (meanfield.py:895-990; eq. 14).  As in the reference the argument ``y`` is NOT what
        enters: the residual is taken against the raw data ``self.y`` (:940, quirk Q3), and of the covariances only the
        diagonals (:956-957).  Computed by the kernel of the device's own ELBO assembly from the state
        ``(mu_f, mu_w)`` / ``(diag Sigma_f, diag Sigma_w)`` and the jitters."""
        q, p, N = self.q, self.p, self.N
        sigma_f = np.asarray(sigma_f, dtype=float).reshape(q, N, N)
        sigma_w = np.asarray(sigma_w, dtype=float).reshape(q, p, N, N)
        mu = np.concatenate((np.reshape(mu_f, (1, q, N)), np.reshape(mu_w, (p, q, N))))
        var = np.empty((p + 1, q, N))
        for j in range(q):
            var[0, j] = np.diag(sigma_f[j])
            for i in range(p):
                var[1 + i, j] = np.diag(sigma_w[j, i])
        ctx = self._backend()
        ctx.set_jitters(np.sqrt(np.asarray(jitt2, dtype=float)))
        ctx.set_muvar(mu, var)
        return np.float64(ctx.expected_loglike())

    def nELBO(self, parameters, max_iter=None):
        """ Negative ELBO at `parameters` (warm-started, meanfield.py:1095-1111) """
        assert self._components_set, _NOT_SET
        self.set_parameters(parameters)
        start = time_module.time()
        elbo, _, _, _ = self.ELBOcalc(self.nodes, self.weights, self.means,
                                      self.jitters, max_iter=max_iter,
                                      mu='previous', var='previous')
        took = 1e3 * (time_module.time() - start)
        print(f'ELBO={elbo:7.2f} (took {took:5.2f} ms)' + 20 * ' ', end='\r', flush=True)
        return -elbo

    def optimize(self, vars=None, **kwargs):
        """
        Maximise the ELBO over the free parameters with scipy.optimize.minimize
        (Nelder-Mead unless `method` is given).  `vars`: 'name' optimises only
        that parameter, '-name' all but it, a list optimises those named.
        ``jac=True`` hands scipy the gradient too (``nELBO_and_grad``; not in the reference, whose optimiser is
        derivative-free).  NOTE that the objective is then a different one: the negative ELBO after a FIXED number of
        forced sweeps (``sweeps=``, default 40) from the state this call starts at, a smooth function of the parameters
        -- ``optimize()`` itself minimises ``nELBO``, the warm-started ELBOcalc under the reference's 1e-3 stop rule,
        which jumps whenever the trip count changes.  The result says so: ``res.objective`` names what ``res.fun`` is,
        and ``res.fun_reference`` is ``nELBO(res.x)``, the reference's objective at the point found.
        """
        from scipy.optimize import minimize
        self._select_vars(vars)
        kwargs.setdefault('method', 'Nelder-Mead')
        # jac=True (not in the reference, which is derivative-free): the analytic gradient of grad_ELBO,
        # e.g. optimize(method='L-BFGS-B', jac=True)
        # ... over a SMOOTH objective: the ELBO after a fixed number of forced sweeps (`sweeps=`, default 40) from the
        # state this call starts at -- the reference's own objective (ELBOcalc under its 1e-3 stop rule, warm-started
        # from the previous evaluation) jumps by 1e-3 relative whenever the trip count changes, which a line search
        # cannot work with
        if kwargs.get('jac') is True:
            sweeps = int(kwargs.pop('sweeps', 40))
            nodes, weights, means, jitters = self._get_components()
            start = (self._mu, self._var) if self._mu is not None else self._initMuVar(nodes, weights, jitters)
            start = (np.array(start[0], dtype=float), np.array(start[1], dtype=float))
            fun = lambda x: self.nELBO_and_grad(x, sweeps=sweeps, start=start)
        else:
            fun = self.nELBO
        smooth = kwargs.get('jac') is True
        res = minimize(fun, self.get_parameters(), **kwargs)
        self.set_parameters(res.x)
        if smooth:
            res.objective = '-ELBO after %d forced sweeps from the state optimize() started at' % sweeps
            res.fun_reference = self.nELBO(res.x)
        else:
            res.objective = 'nELBO: -ELBO of the warm-started ELBOcalc under the 1e-3 stop rule (meanfield.py:1095-1111)'
            res.fun_reference = res.fun
        return res


    # ------------------------------------------------------------ prediction
    def _kss_diagonal(self, kernel, tstar):
        """diag of ``_tinyNuggetKMatrix(kernel, tstar)`` (what _gp.GP.prediction takes of K**, _gp.py:130-137) without
        forming the n* x n* matrix: point by point through the kernel's own ``__call__`` -- ``kernel(0)`` for the
        one-argument kernels (plus the 1.25e-12 nugget), ``kernel(t, t)`` for the two-argument ones."""
        tstar = np.atleast_1d(np.asarray(tstar, dtype=float))
        if isinstance(kernel, _TWO_ARGUMENT):                  # (no nugget for these: meanfield.py:447-450)
            return np.array([np.ravel(kernel(np.array([[ts]]), np.array([[ts]])))[0] for ts in tstar], dtype=float)
        return np.full(tstar.size, float(np.ravel(kernel(np.zeros((1, 1))))[0])) + _TINY_NUGGET

    def _Prediction(self, nodes=None, weights=None, means=None, jitters=None,
                    tstar=None, mu=None, var=None, separate=False):
        """
        GPRN predictive mean and variance at `tstar` (default: the data times)
        from variational means/variances `mu`, `var` (default: the converged
        ones of the last ELBOcalc, else the `_initMuVar` start point) --
        meanfield.py:1289-1381.  Every latent GP's conditional mean/variance
        (the reference's `_gp.GP.prediction`, one Cholesky + N* solves each) is
        computed on the GPU (`gprn_predict`; on a sharded object by the latent GP's
        owner, every rank receiving all rows); the O(p q N*) combination below is
        the reference's, including the jitter added once per node.

        Returns (mean (N*, p), variance (N*, p)) and, with `separate`, the
        object array [node means (q, N*), weight means (q*p, N*)].
        """
        nodes, weights, means, jitters = self._get_components(nodes, weights, means, jitters)
        tstar = self.time if tstar is None else np.atleast_1d(np.asarray(tstar, dtype=float))
        if mu is None and var is None:
            if self._mu is None and self._var is None:
                mu, var = self._initMuVar(nodes, weights, jitters)
            else:
                mu, var = self._mu, self._var

        specs = [self._kernel_spec(k) for k in chain(nodes, weights)]
        ctx = self._backend()
        key = tuple(self._spec_key(sp) for sp in specs)
        if key != self._prior_key:
            for gp, sp in enumerate(specs):
                self._send_spec(ctx, gp, sp)
            self._prior_key = None             # the priors must be refactored before the next sweep
        # a user-defined covFunction has no device form of its cross-covariance: its three matrices are
        # evaluated here, as the reference evaluates every kernel's (_gp.py:40-63, meanfield.py:436-471),
        # and handed over; the factorisation and the solves stay on the GPU (owner rank only when sharded)
        data_t = np.asarray(self.time, dtype=float)
        for gp, (sp, kernel) in enumerate(zip(specs, chain(nodes, weights))):
            if sp[0] != 'host' or ctx.owner_of(gp) != ctx.rank:
                continue
            ctx.predict_upload(gp, self._tinyNuggetKMatrix(kernel, data_t),
                               self._predictKMatrix(kernel, tstar), self._kss_diagonal(kernel, tstar))
        ctx.set_muvar(np.asarray(mu, dtype=float), np.asarray(var, dtype=float))
        gmean, gvar, info = ctx.predict(tstar)
        self.last_info = info

        q, p = self.q, self.p
        nPred, nVar = gmean[:q], gvar[:q]
        wPred, wVar = gmean[q:], gvar[q:]
        wP, wV = wPred.reshape(q, p, tstar.size), wVar.reshape(q, p, tstar.size)
        meanVal = np.array(np.array_split(self._mean(means, tstar), p))
        jitt2 = np.array(jitters)**2
        predictives = np.zeros((tstar.size, p))
        predictivesVar = np.zeros((tstar.size, p))
        for i in range(p):
            predictives[:, i] += meanVal[i]
            for j in range(q):
                predictives[:, i] += nPred[j] * wP[j, i]
                predictivesVar[:, i] += wP[j, i] * wP[j, i] * nVar[j] \
                    + wV[j, i] * (nVar[j] + nPred[j] * nPred[j]) + jitt2[i]
        if separate:
            return predictives, predictivesVar, np.array([nPred, wPred], dtype=object)
        return predictives, predictivesVar

    def predict(self, tstar=None, nn=1000):
        """
        GPRN prediction at `tstar`, or on `nn` points spanning the data padded by
        20 % on each side (meanfield.py:1383-1400).  Returns (tstar, mean,
        standard deviation, [node means, weight means]).
        """
        if tstar is None:
            lo, hi = self.time.min(), self.time.max()
            span = np.ptp(self.time)
            tstar = np.linspace(lo - 0.2 * span, hi + 0.2 * span, nn)
        mean, variance, parts = self._Prediction(tstar=tstar, separate=True)
        return tstar, mean, np.sqrt(variance), parts

    #: largest N for which nELBO_batch / mcmc(batch=True) evaluate side by side on one GPU (beyond it a single evaluation
    #: already fills the device, and a chunk of evaluations would be a few matrices)
    batch_max_N = 2048
    _batch_last_done = -1            # position (in the last side-by-side list) of the evaluation whose state was kept

    def nELBO_batch(self, parameter_sets, max_iter=None, pool=None, batch=True):
        """
        ``nELBO`` for several free-parameter vectors: ``[nELBO(p) for p in
        parameter_sets]``.  With ``pool`` (``sharding.EvalPool``; every rank holds
        this same problem on its own GPU) the vectors are split over the ranks and
        every rank returns the full list.  Not in the reference, which evaluates
        optimiser populations and emcee walkers one by one (meanfield.py:1222-1260).
        The parameters of ``self`` end up at the last vector this rank evaluated.  With a pool AND the side-by-side form
        below, each rank evaluates its share side by side and the state every rank keeps is the same (that of the last
        evaluation of the whole list that converged): the values do not depend on the number of ranks.

        Without a pool, a problem whose kernels all have device programs evaluates the whole list SIDE BY SIDE on the GPU
        (``gprn_elbocalc_batch``): every evaluation with its own covariance matrices, state, loop and stop rule, all of
        them starting from the state the object holds (``nELBO``'s warm start) -- where one evaluation after the other
        starts each from its predecessor's result, which moves the values within the stop rule's 1e-3.  One-tile problems
        (N <= 128) run a half-sweep of all evaluations as ONE launch; larger ones (up to ``batch_max_N`` points) go
        through the large problems' launch schedule with its batch dimension = evaluations x latent GPs, in chunks that
        fit the library's memory budget.  ``batch=False`` forces the one-by-one form.
        """
        assert self._components_set, _NOT_SET
        sets = [np.array(x, dtype=float) for x in parameter_sets]
        if pool is None and batch and len(sets) > 1:
            out = self._nELBO_batch_device(sets, max_iter)
            if out is not None:
                return out
        if pool is not None and batch and sets and hasattr(pool, 'map_lists') and self._batchable():
            return self._nELBO_batch_pool(sets, max_iter, pool)
        f = lambda x: float(self.nELBO(x, max_iter=max_iter))
        return list(map(f, sets)) if pool is None else pool.map(f, sets)

    def _batchable(self):
        """Whether ``_nELBO_batch_device`` applies to this object at its current components -- a property of the problem,
        the same on every rank of a pool (so that all of them take the same branch, collectives included)."""
        if self._comm is not None or self.N > self.batch_max_N:
            return False
        nodes, weights, _, _ = self._get_components()
        return all(self._kernel_spec(k)[0] == 'device' for k in chain(nodes, weights))

    def _nELBO_batch_pool(self, sets, max_iter, pool):
        """``nELBO_batch`` over the GPUs of a node: rank r evaluates ``sets[r::world]`` SIDE BY SIDE on its own GPU, all
        of them from the state the object holds on entry -- the same on every rank, because the state it holds on exit
        is again the same on every rank: that of the last evaluation OF THE WHOLE LIST whose loop converged
        (``nELBO_batch``'s rule), handed round by its owner.  The values therefore do not depend on the number of
        ranks (to the rounding of a batch's size-dependent launch shapes)."""
        n = len(sets)
        mine = list(range(pool.rank, n, pool.world))
        last = {'key': -1}

        def share(xs):
            out = self._nELBO_batch_device(xs, max_iter)
            if out is None:                                    # (vectors that change a kernel expression's shape)
                # one by one: the state to hand round is that of the last evaluation whose loop CONVERGED -- ELBOcalc
                # stores a state only then (a new array object) -- and only such a state is offered: with nothing stored
                # the ranks would meet in the collective with buffers of different sizes
                out = []
                for i, x in enumerate(xs):
                    before = self._mu
                    out.append(float(self.nELBO(x, max_iter=max_iter)))
                    if self._mu is not None and self._mu is not before:
                        last['key'] = mine[i]
            elif self._batch_last_done >= 0:
                last['key'] = mine[self._batch_last_done]
            return out

        mu0, var0 = self._mu, self._var
        vals = pool.map_lists(share, sets)
        shape = (self.p + 1, self.q, self.N)
        offer = [np.reshape(self._mu, shape), np.reshape(self._var, shape)] if last['key'] >= 0 else [np.zeros(shape), np.zeros(shape)]
        got = pool.take_from_highest(last['key'], offer)
        if got is not None:
            self._mu, self._var = got
        else:
            self._mu, self._var = mu0, var0
        self.set_parameters(sets[-1])
        return vals

    def _nELBO_batch_device(self, sets, max_iter):
        """``nELBO_batch`` through ``gprn_elbocalc_batch``, or None where that does not apply (larger problems, sharded
        objects, user-defined kernels, kernel expressions that change shape from one vector to the next)."""
        if self._comm is not None or self.N > self.batch_max_N:
            return None
        ctx = self._backend()
        max_iter = 10000 if max_iter is None else int(max_iter)
        y_raw = np.concatenate(self.y)
        start = time_module.time()
        B = len(sets)
        # (set_parameters takes full-length vectors too, meanfield.py:223-259: the fast layout below wants them all alike)
        n_free, n_all = int((~self.frozen_mask).sum()), int(self.frozen_mask.size)
        if any(x.ndim != 1 or x.size not in (n_free, n_all) for x in sets):
            return None                                        # (the one-by-one form raises the reference's ValueError)
        if n_free != n_all and any(x.size == n_all for x in sets):
            free_ = ~self.frozen_mask
            sets = [x[free_] if x.size == n_all else x for x in sets]
        # the first vector the ordinary way: it tells what the programs are and whether the rest can be laid out in one go
        self.set_parameters(sets[0])
        nodes, weights, means, jitters = self._get_components()
        kernels = list(chain(nodes, weights))
        specs = [self._kernel_spec(k) for k in kernels]
        if any(sp[0] != 'device' for sp in specs):
            return None
        shape_ref = tuple((sp[1], sp[3]) for sp in specs)
        for gp, sp in enumerate(specs):                        # the programs the library substitutes the parameters into
            self._send_spec(ctx, gp, sp)
        self._prior_key = None                                 # (the object's own factors are stale now)
        n_k = sum(k.pars.size for k in kernels)
        plain_kernels = all(sp[2].size == k.pars.size and np.array_equal(sp[2], k.pars) for sp, k in zip(specs, kernels))
        plain_means = all(m_ is None or type(m_) is meanfunc.Constant for m_ in means)
        if plain_kernels and plain_means and self._mu is not None:
            # every program's parameters ARE its kernel's, the means are constants and the start is the stored state:
            # the B problems are slices of the B full parameter vectors (nodes, weights, means, jitters: meanfield.py:193-202)
            full = np.tile(self.get_parameters(include_frozen=True), (B, 1))
            free = ~self.frozen_mask
            full[:, free] = np.array(sets)
            kp = full[:, :n_k]
            n_m = sum(0 if m_ is None else 1 for m_ in means)
            yr = np.tile(y_raw, (B, 1))
            col = n_k
            for i, m_ in enumerate(means):
                if m_ is not None:
                    yr[:, i * self.N:(i + 1) * self.N] -= full[:, col:col + 1]
                    col += 1
            jt = full[:, n_k + n_m:]
            m0 = np.tile(np.ravel(self._mu), (B, 1))
            v0 = np.tile(np.ravel(self._var), (B, 1))
            self.set_parameters(sets[-1])
        else:
            kp, yr, jt, m0, v0 = [], [], [], [], []
            for i, x in enumerate(sets):
                if i:
                    self.set_parameters(x)
                    nodes, weights, means, jitters = self._get_components()
                    specs = [self._kernel_spec(k) for k in chain(nodes, weights)]
                    if any(sp[0] != 'device' for sp in specs) or tuple((sp[1], sp[3]) for sp in specs) != shape_ref:
                        return None
                kp.append(np.concatenate([sp[2] for sp in specs]))
                yr.append(y_raw - self._mean(means))
                jt.append(np.asarray(jitters, dtype=float))
                if self._mu is not None:
                    mu, var = self._mu, self._var
                else:
                    mu, var = self._initMuVar(nodes, weights, jitters)
                m0.append(np.ravel(mu))
                v0.append(np.ravel(var))
        res = ctx.elbocalc_batch(np.array(kp), np.array(yr), np.array(jt), np.array(m0), np.array(v0), max_iter,
                                 want_state=True)
        if res is None:
            return None
        elbo, iters, conv, info, mu_f, var_f = res
        self.last_info = int(info[np.flatnonzero(info)[0]]) if np.any(info) else 0
        done = np.flatnonzero(conv)
        self._batch_last_done = int(done[-1]) if done.size else -1
        if done.size:                                      # the warm start of whatever comes next (meanfield.py:644-646)
            self._mu, self._var = mu_f[done[-1]], var_f[done[-1]]
        took = 1e3 * (time_module.time() - start)
        print(f'{len(sets)} ELBO evaluations side by side (took {took:5.2f} ms)' + 20 * ' ', end='\r', flush=True)
        return [float(-e) for e in elbo]

    # ------------------------------------------------------------ gradients
    def grad_ELBO(self, mean_sweeps=8, mean_start=None, total=False):
        """
        Gradient of the ELBO with respect to ALL parameters (the order of ``get_parameters(
        include_frozen=True)``: nodes, weights, means, jitters) at the current variational state.

        Not in the reference, whose optimiser is derivative-free (meanfield.py:1149-1150).
        One more committed sweep is run from the stored state with the explicit covariances kept; the
        gradient is the partial derivative of THAT sweep's ELBO (returned with it) at fixed variational
        means and covariances -- what the envelope theorem makes the total derivative at a converged state:

        * kernel hyper-parameters enter through the expected log prior only (meanfield.py:992-1067):
          ``dELBO/dtheta = 1/(2q) < K^-1 S K^-1 + a a^T - K^-1 , dK/dtheta >`` with ``a = K^-1 m`` and the
          (S, m) the reference pairs with that kernel -- quirks included: node j meets the cumulative
          ``Sigma_f0 + ... + Sigma_fj`` (Q1), weight (j, i) the raw-reshape row of ``mu_w`` (Q2).  The
          N^3 work (K^-1, K^-1 S K^-1) runs on the GPU (``gprn_grad_matrices``); ``dK/dtheta`` comes from
          ``covFunction._dk_dpars`` (closed forms for SquaredExponential, Periodic, QuasiPeriodic; central
          differences of ``kernel(r)`` otherwise, user kernels included).
        * jitters enter through the expected log likelihood (meanfield.py:895-990), in closed form.
        * mean-function parameters: at a fixed variational state, zero -- the reference's likelihood term reads
          the RAW data (quirk Q3), so the reported ELBO does not see them.  They act through the residual
          ``y - mean`` that the coordinate-ascent update reads (meanfield.py:623-624, 765-792), i.e. by moving the
          state the sweeps converge to; and because the update maximises a bound on the mean-subtracted data
          while the ELBO is evaluated on the raw data, that state is not stationary for the reported ELBO and the
          envelope theorem does not make the effect vanish.  With ``mean_sweeps > 0`` (default 8) these few
          entries are therefore central differences (relative step 1e-4) of the ELBO after ``mean_sweeps`` forced
          sweeps from the stored state with the mean parameter moved, everything else as it is: two short device
          runs per mean parameter, priors untouched (``mean_start``: run them from this ``(mu, var)`` instead of the
          stored state).  ``mean_sweeps=0`` gives the partial derivative (zeros).

        ``total=True``: the gradient of what the sweeps CONVERGE to, for every parameter.  With zero mean functions that
        is the fixed-state gradient above (the envelope theorem holds: checked to 1e-5 in the tests).  With non-zero mean
        functions it is not -- the fixed-state entries of the kernel parameters miss it by 2-85 % and one jitter even has
        the wrong sign at the test problem -- so then EVERY free parameter gets the finite-difference treatment of the
        mean-function parameters: central differences of the ELBO after ``mean_sweeps`` forced sweeps (two set-ups and
        runs per kernel parameter: this is for small problems and few parameters).

        Returns ``(ELBO, gradient)``.  Unsharded problems only.
        """
        assert self._components_set, _NOT_SET
        if self._mu is None:
            self.ELBOcalc()
        nodes, weights, means, jitters = self._get_components()
        ctx = self._setup_device(nodes, weights, means, jitters)
        mu_in, var_in = np.array(self._mu, dtype=float), np.array(self._var, dtype=float)
        ctx.set_muvar(mu_in, var_in)
        ctx.keep_sigma(True)
        try:
            elbo, _, info = ctx.sweep(1, commit=True)
            mu, var = ctx.get_muvar()
            grads = self._grad_from_state(nodes, weights, means, jitters, mu, var, ctx.grad_matrices,
                                          device=ctx.grad_kernel)
        finally:
            ctx.keep_sigma(False)
        self._mu, self._var = mu, var
        self.last_info = info
        grads = np.array(grads)
        n_k = sum(k.pars.size for k in chain(nodes, weights))
        n_m = sum(0 if m_ is None else int(m_._parsize) for m_ in means)
        if total and mean_sweeps > 0 and np.any(self._mean(means) != 0.0):
            m0, v0 = (mu_in, var_in) if mean_start is None else mean_start
            grads[:] = self._mean_parameter_differences(0, grads.size, m0, v0, int(mean_sweeps), (mu, var), rel_step=1e-5)
        elif mean_sweeps > 0:
            if n_m:
                m0, v0 = (mu_in, var_in) if mean_start is None else mean_start
                grads[n_k:n_k + n_m] = self._mean_parameter_differences(n_k, n_m, m0, v0, int(mean_sweeps), (mu, var))
        return float(elbo[0]), grads

    def _mean_parameter_differences(self, first, count, mu, var, n_sweeps, restore, rel_step=1e-4):
        """d/dtheta of the ELBO after `n_sweeps` forced sweeps from the state (mu, var), for the `count`
        mean-function parameters that start at position `first` of the full parameter vector: central
        differences, the device's priors untouched (only ``y - mean`` changes); the device state ends as `restore`."""
        full = self.get_parameters(include_frozen=True).copy()
        out = np.zeros(count)
        try:
            for n in range(count):
                if self.frozen_mask[first + n]:
                    continue                                      # not a variable of anybody's optimisation
                v = full[first + n]
                h = rel_step * max(1.0, abs(v))
                e = []
                for sign in (1.0, -1.0):
                    x = full.copy()
                    x[first + n] = v + sign * h
                    self.set_parameters(x)
                    nodes, weights, means, jitters = self._get_components()
                    ctx = self._setup_device(nodes, weights, means, jitters)
                    ctx.set_muvar(mu, var)
                    e.append(ctx.sweep(n_sweeps, commit=False)[0][-1])
                out[n] = (e[0] - e[1]) / (2 * h) if np.all(np.isfinite(e)) else 0.0
        finally:
            self.set_parameters(full)
            nodes, weights, means, jitters = self._get_components()
            self._setup_device(nodes, weights, means, jitters).set_muvar(*restore)
        return out

    def _grad_from_state(self, nodes, weights, means, jitters, mu, var, matrices, device=None):
        """The O(N^2) and O(pqN) part of grad_ELBO: `matrices(gp)` returns ``(K^-1, K^-1 S K^-1)`` of latent GP
        `gp` (the GPU's ``gprn_grad_matrices``; a NumPy stand-in in the CPU tests).  `device(gp, m, n)`, when
        given, is tried first: the whole contraction on the GPU (``gprn_grad_kernel``: closed-form or
        central-difference dK/dtheta of the kernel's device program), None for kernels without one."""
        t = np.asarray(self.time, dtype=float)
        r = t[:, None] - t[None, :]
        q, p, N = self.q, self.p, self.N
        m_scr = mu[1:].reshape(q, p, N)                      # quirk Q2 (meanfield.py:1021)
        grads = []
        for gp, kernel in enumerate(chain(nodes, weights)):
            if gp < q:
                m = mu[0, gp]
            else:
                jj, ii = divmod(gp - q, p)
                m = m_scr[jj, ii]
            if device is not None and kernel._device_program() is not None:
                on_device = device(gp, m, kernel.pars.size)
                if on_device is not None:
                    grads += [float(v) / q for v in on_device]
                    continue
            Kinv, P = matrices(gp)
            a = Kinv @ m
            G = 0.5 * (P - Kinv + np.outer(a, a)) / q        # ELBO = (...) / q, meanfield.py:709
            if isinstance(kernel, _TWO_ARGUMENT):
                keep = kernel.pars.copy()
                dks = []
                for i, v in enumerate(keep):                  # kernel(t_i, t_j): differences only
                    h = 1e-6 * max(1.0, abs(v))
                    kernel.pars = keep.copy(); kernel.pars[i] = v + h
                    up = kernel(t[:, None], t[None, :])
                    kernel.pars = keep.copy(); kernel.pars[i] = v - h
                    dks.append((up - kernel(t[:, None], t[None, :])) / (2 * h))
                kernel.pars = keep
            else:
                dks = kernel._dk_dpars(r)
            grads += [float(np.sum(G * dk)) for dk in dks]
        grads += [0.0] * sum(0 if m_ is None else int(m_._parsize) for m_ in means)
        # jitters: LogL = -1/2 sum [log(2 pi v) + ((Y - fit)^2 + A) / v],  v = jitter^2 + yerr^2
        variance = np.asarray(jitters, dtype=float)[:, None]**2 + self.yerr2
        fit = np.einsum('iqn,qn->in', mu[1:], mu[0])
        A = np.zeros((p, N))
        for i in range(p):
            for j in range(q):
                A[i] += var[0, j] * mu[1 + i, j]**2 + var[1 + i, j] * mu[0, j]**2 + var[0, j] * var[1 + i, j]
        dv = -0.5 * (1.0 / variance - ((self.y - fit)**2 + A) / variance**2)
        grads += [float(np.sum(dv[i]) * 2 * jitters[i]) / q for i in range(p)]
        return grads

    def nELBO_and_grad(self, parameters, max_iter=None, sweeps=None, start=None):
        """``(-ELBO, -dELBO/dparameters)`` over the FREE parameters, for gradient-based optimisers.  Default:
        ``nELBO(parameters)`` (warm-started ELBOcalc, as the reference's objective), then ``grad_ELBO``.  With
        ``sweeps`` (and a start state ``(mu, var)``): the ELBO after exactly that many forced sweeps from ``start``
        plus the one ``grad_ELBO`` adds -- a deterministic, smooth function of the parameters."""
        if sweeps is None:
            self.nELBO(parameters, max_iter=max_iter)
            elbo, grad = self.grad_ELBO()
            return -elbo, -grad[~self.frozen_mask]
        assert self._components_set, _NOT_SET
        self.set_parameters(np.array(parameters, dtype=float))
        nodes, weights, means, jitters = self._get_components()
        if start is None:
            start = self._initMuVar(nodes, weights, jitters)
        ctx = self._setup_device(nodes, weights, means, jitters)
        ctx.set_muvar(np.asarray(start[0], dtype=float), np.asarray(start[1], dtype=float))
        _, _, info = ctx.sweep(int(sweeps), commit=True)
        self.last_info = info
        self._mu, self._var = ctx.get_muvar()
        elbo, grad = self.grad_ELBO(mean_sweeps=int(sweeps) + 1, mean_start=start, total=True)
        if not np.isfinite(elbo):
            return np.inf, np.zeros(int((~self.frozen_mask).sum()))
        return -elbo, -grad[~self.frozen_mask]

    def mcmc(self, priors, p0=None, vars=None, niter=500, **kwargs):
        """
        Sample the posterior of the free parameters with emcee, the ELBO (100
        sweeps at most, warm-started) standing in for the marginal likelihood
        (meanfield.py:1154-1286).  `priors`: dict name -> frozen scipy.stats
        distribution.  emcee is imported here, not at package import; without it
        this raises ImportError.  Returns the sampler.

        Same sequence as the reference: walkers from the priors (or an ellipsoid around `p0`), one
        evaluation of every initial walker (it also moves the warm-start state), a sampler over an HDF
        backend file ``gprn.h5`` that is reset on every call, convergence from the autocorrelation time
        every ten steps.  Beyond the reference: keyword arguments go on to ``emcee.EnsembleSampler`` (the
        reference accepts and drops them), so ``pool=sharding.EvalPool()`` spreads the walkers over the
        GPUs of a node; ``backend=`` replaces the HDF file, and without h5py emcee's in-memory backend is used.
        ``batch=True`` hands emcee a VECTORISED log-probability: the walkers of a half-step are evaluated side by side
        on the GPU (``nELBO_batch``), each from the SAME warm-start state -- the one the object holds when the half-step
        begins -- instead of from its predecessor's converged state (meanfield.py:1102-1104).  What that changes: an
        evaluation is 100 sweeps at most under the 1e-3 stop rule, which fires when progress is slow, not when the fixed
        point is near, so its value depends on where its loop started.  Measured on 40 prior draws of a small problem
        (``tests/test_parity_gpu.py::test_mcmc_with_the_walkers_side_by_side``): side by side against chained, median
        4e-4 relative, nine in ten within 4e-3, worst 0.17 -- and the reference's own chaining differs from itself by the
        same amounts when the same vectors are evaluated in the opposite order.  The batched log-probabilities are
        therefore as good an estimate of the objective as the reference's; the chain is not the reference's walker for
        walker (one accept / reject decision that falls inside that spread sends the ensembles apart for good: a moved
        walker changes every later proposal), the posterior it samples is.
        """
        assert self._components_set, _NOT_SET
        from emcee import EnsembleSampler, backends
        self._select_vars(vars)
        names = np.array(list(self.parameters_dict.keys()))[~self.frozen_mask]

        def draw():
            return np.array([priors[n].rvs() for n in names])

        def logprior(x):
            return float(sum(priors[n].logpdf(v) for v, n in zip(x, names)))

        def logposterior(x):
            lp = logprior(x)
            if np.isneginf(lp):
                return -np.inf, -np.inf
            elbo = -self.nELBO(x, max_iter=100)
            return lp + elbo, elbo

        def logposterior_batch(X):
            X = np.atleast_2d(X)
            # (the priors of all walkers at once, name by name in logprior's order: the same sums, 1 / nwalkers of the
            # scipy.stats calls -- at 40 walkers x 20 parameters those calls outweighed the side-by-side evaluations)
            lp = np.zeros(X.shape[0])
            for k, n in enumerate(names):
                lp = lp + np.asarray(priors[n].logpdf(X[:, k]), dtype=float)
            out = np.full((X.shape[0], 2), -np.inf)
            ok = np.flatnonzero(~np.isneginf(lp))
            if ok.size:
                elbo = -np.array(self.nELBO_batch([X[i] for i in ok], max_iter=100, pool=batch_pool))
                out[ok, 0] = lp[ok] + elbo
                out[ok, 1] = elbo
            return out

        batch = bool(kwargs.pop('batch', False))
        # (a vectorised log-probability never reaches emcee's pool: with both, each rank of the pool evaluates its share
        # of a half-step's walkers side by side -- nELBO_batch(pool=...))
        pool_rank = getattr(kwargs.get('pool'), 'rank', 0)
        batch_pool = kwargs.pop('pool', None) if batch else None

        ndim = len(names)
        nwalkers = 2 * ndim
        print(f'Setting up sampler (parameters: {ndim}, walkers: {nwalkers})')
        if p0 is None:
            p0 = np.array([draw() for _ in range(nwalkers)])
        else:
            from emcee.utils import sample_ellipsoid
            sigma = []
            for n in names:
                try:
                    sigma.append(priors[n].std())
                except TypeError:
                    sigma.append(priors[n].std)
            p0 = sample_ellipsoid(p0, np.diag(sigma) / 100, size=nwalkers)
            for i, x in enumerate(p0):
                if np.isneginf(logprior(x)):
                    p0[i] = draw()
        print('initial values for parameters are set')
        start = time_module.time()
        _ = logposterior_batch(p0) if batch else [logposterior(x) for x in p0]
        print()
        print(f'evaluation for initial values took {time_module.time() - start:.0f} sec')
        print('- adjust your expectations accordingly')

        # the reference's HDF5 file (meanfield.py:1262-1263) -- written by ONE process: with an SPMD pool
        # (sharding.EvalPool) every rank runs this same method, and N writers resetting one file end in an
        # h5py lock error or a corrupt file; the other ranks keep emcee's in-memory backend
        if 'backend' not in kwargs and pool_rank == 0:
            try:
                be = backends.HDFBackend('gprn.h5')
                be.reset(nwalkers, ndim)
                kwargs['backend'] = be
            except (ImportError, OSError):     # h5py missing or the file is locked: emcee's in-memory backend
                pass
        if batch:
            kwargs['vectorize'] = True
        sampler = EnsembleSampler(nwalkers, ndim, logposterior_batch if batch else logposterior, **kwargs)
        old_tau = np.inf
        for sample in sampler.sample(p0, iterations=niter, progress=True):
            if sampler.iteration % 10 == 0:
                print(sample.log_prob.max())
            if sampler.iteration % 10:
                continue
            tau = sampler.get_autocorr_time(tol=0)
            converged = np.all(tau * 100 < sampler.iteration)
            converged &= np.all(np.abs(old_tau - tau) / tau < 0.01)
            if converged:
                print('MCMC converged!')
                break
            old_tau = tau
        return sampler

    def _select_vars(self, vars):
        """ Freeze/thaw by the `vars` convention shared by optimize and mcmc:
        'name' frees only that parameter, '-name' all but it, a list frees those. """
        if vars is None:
            return
        if isinstance(vars, str):
            if '-' in vars:
                self.thaw_parameter(name='*')
                self.freeze_parameter(name=vars.replace('-', ''))
            else:
                self.freeze_parameter(name='*')
                self.thaw_parameter(name=vars)
        elif isinstance(vars, list):
            self.freeze_parameter(name='*')
            for name in vars:
                self.thaw_parameter(name=name)
        else:
            raise ValueError(f'`vars` should be str or list, got {type(vars)}')

