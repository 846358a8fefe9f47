"""In-memory stand-in for emcee.backends (TEST INFRASTRUCTURE): the reference resets an HDF file per run."""


class Backend:
    def __init__(self, *a, **k):
        self.reset_args = None

    def reset(self, nwalkers, ndim):
        self.reset_args = (nwalkers, ndim)


class HDFBackend(Backend):
    def __init__(self, filename, *a, **k):
        super().__init__()
        self.filename = filename
