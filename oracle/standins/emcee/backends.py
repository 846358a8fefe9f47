class HDFBackend:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("emcee stand-in")
