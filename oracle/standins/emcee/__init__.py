"""Placeholder for the absent `emcee` package (never called on the ELBO path)."""


class EnsembleSampler:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("emcee stand-in: MCMC is outside the golden-vector scope")


from . import backends, utils  # noqa: E402,F401
