"""Sharding of the q + q*p latent GPs over the GPUs of one node.

Not in the reference (it is single-process NumPy).  The latent GPs of one
half-sweep are independent (meanfield.py:769-792 nodes, :846-865 weights), so
each rank factors and updates only the GPs it owns; per sweep the owners then
broadcast their O(N) rows of mu/var (one grouped RCCL call per half-sweep) and
one all-reduce sums the per-GP ELBO scalars -- see DESIGN.md §5.  No N x N
matrix ever crosses xGMI.

One process per GPU (``python -m torch.distributed.run`` or any launcher that
sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT); the ncclUniqueId travels
through a file in /tmp keyed by the launcher's pid, so no torch import is
needed here.

A second, coarser way to use several GPUs is `EvalPool`: every rank keeps the WHOLE
problem on its own GPU and the ranks split a list of independent evaluations
(parameter vectors of an optimiser population, emcee walkers -- the reference
evaluates those one by one, meanfield.py:1222-1260); no N x N data and no per-sweep
exchange at all, one small all-reduce per batch.
"""
import os
import time

_generation = 0
_START = time.time()


def owners(p, q, world):
    """Rank that owns each latent GP, in the library's index order: nodes
    0..q-1, then weights q + j*p + i.  Nodes go round-robin from rank 0, weights
    continue round-robin after them, which balances each half-sweep on its own
    (the two half-sweeps are sequential)."""
    out = [j % world for j in range(q)]
    out += [(q + k) % world for k in range(q * p)]
    return out


def local_gps(p, q, world, rank):
    own = owners(p, q, world)
    nodes = [g for g in range(q) if own[g] == rank]
    weights = [g for g in range(q, q + q * p) if own[g] == rank]
    return nodes, weights


def helper_inverses(p, q, world, rank):
    """Nodes j whose explicit K_j^-1 this rank must hold for the reference's
    cumulative-trace quirk (meanfield.py:1039-1041: node j's trace term sums the
    Sigma of every node k <= j): all j > the smallest locally owned node."""
    nodes, _ = local_gps(p, q, world, rank)
    if q < 2 or not nodes:
        return []
    return list(range(nodes[0] + 1, q))


def _rendezvous_dir():
    """A directory only this user can write to: $XDG_RUNTIME_DIR when it exists, else /tmp/gprn-<uid> (0700)."""
    base = os.environ.get('XDG_RUNTIME_DIR')
    if base and os.path.isdir(base) and os.access(base, os.W_OK):
        d = os.path.join(base, 'gprn')
    else:
        d = os.path.join('/tmp', 'gprn-%d' % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f'{d} is not a private directory of this user')
    return d


class Comm:
    """World description + ncclUniqueId rendezvous for one communicator.

    The id travels through a file in a private per-user directory, named after the launch (the
    launcher's pid, or GPRN_LAUNCH_TAG when the launcher sets it), the rendezvous port, the world size
    and a per-process generation count.  Rank 0 removes anything stale under that name, creates the file
    with O_EXCL and mode 0600 and renames it into place; the other ranks accept only a 128-byte file
    of their own user that is at most minutes older than their own start.  ``done()`` removes it once the communicator exists.
    """

    def __init__(self, world=None, rank=None, local_rank=None, tag=None):
        env = os.environ
        self.world = int(env.get('WORLD_SIZE', 1)) if world is None else int(world)
        self.rank = int(env.get('RANK', 0)) if rank is None else int(rank)
        self.local_rank = int(env.get('LOCAL_RANK', self.rank)) if local_rank is None \
            else int(local_rank)
        global _generation
        _generation += 1
        port = env.get('MASTER_PORT', '0')
        if tag is None:
            tag = env.get('GPRN_LAUNCH_TAG') or os.getppid()
        self._path = None
        self._name = 'uid_%s_%s_%s_%d' % (tag, port, self.world, _generation)
        self._id = None

    def _file(self):
        if self._path is None:
            self._path = os.path.join(_rendezvous_dir(), self._name)
        return self._path

    def unique_id(self, timeout=300.0):
        """Rank 0 creates the id and publishes it atomically; the others wait."""
        if self._id is not None or self.world == 1:
            return self._id
        path = self._file()
        if self.rank == 0:
            from . import _hip
            self._id = _hip.comm_unique_id()
            tmp = path + '.tmp%d' % os.getpid()
            for stale in (path, tmp):
                try:
                    os.unlink(stale)
                except FileNotFoundError:
                    pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, 'wb') as f:
                f.write(self._id)
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while True:
                try:
                    st = os.stat(path)
                    # written during this launch: the ranks of one launch start within seconds of each other, a
                    # left-over of an earlier launch under the same name (same launcher pid, port, world) is far older
                    fresh = st.st_mtime >= _START - 600.0
                    if st.st_uid == os.getuid() and not (st.st_mode & 0o077) and st.st_size == 128 and fresh:
                        with open(path, 'rb') as f:
                            data = f.read()
                        if len(data) == 128:
                            self._id = data
                            break
                except FileNotFoundError:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError(f'no ncclUniqueId at {path} after {timeout}s')
                time.sleep(0.01)
        return self._id

    def done(self):
        """The communicator is up on this rank (its creation is collective: every rank has read the id)."""
        if self.rank == 0 and self._path is not None:
            try:
                os.unlink(self._path)
            except OSError:
                pass

    cleanup = done


class EvalPool:
    """``map`` over the ranks of one node, for independent evaluations.

    SPMD: every rank runs the same script and calls ``pool.map(func, items)`` with
    the same ``items``; rank r evaluates ``items[r::world]`` on its own GPU and one
    all-reduce hands every rank the full result list, in order.  ``func`` must
    return a float or a fixed-length tuple of floats (emcee's log-probability with
    blobs).  This is the ``pool`` protocol of ``emcee.EnsembleSampler`` and of
    ``inference.nELBO_batch``; seed the ranks identically so that they propose the
    same walkers.

    The pool owns a small library context used only for the collective; the
    ``inference`` objects doing the work are ordinary unsharded ones created with
    ``device=pool.device``.
    """

    def __init__(self, comm=None):
        from . import _hip
        self.comm = Comm() if comm is None else comm
        self.world, self.rank = self.comm.world, self.comm.rank
        self.device = self.comm.local_rank % max(1, _hip.device_count())
        self._ctx = _hip.Context(self.device)
        if self.world > 1:
            self._ctx.comm_init(self.world, self.rank, self.comm.unique_id())
            self.comm.done()

    def map(self, func, items):
        items = list(items)
        n = len(items)
        if n == 0:
            return []
        mine = range(self.rank, n, self.world)
        local = {i: func(items[i]) for i in mine}
        if self.world == 1:
            return [local[i] for i in range(n)]
        import numpy as np
        # width of one result: known on the ranks that evaluated something, shared by a max
        first = next(iter(local.values())) if local else 0.0
        scalar = np.ndim(first) == 0
        width = int(self._ctx.barrier_max(1 if scalar else len(first)))
        scalar = bool(self._ctx.barrier_max(0.0 if scalar else 1.0) == 0.0)
        buf = np.zeros((n, width))
        for i, v in local.items():
            buf[i] = np.atleast_1d(np.asarray(v, dtype=float))
        # a sum with zeros elsewhere; -inf and nan survive it
        buf = self._ctx.allreduce_sum(buf).reshape(n, width)
        if scalar:
            return [float(buf[i, 0]) for i in range(n)]
        return [tuple(float(x) for x in buf[i]) for i in range(n)]

    def map_lists(self, func, items):
        """``map`` for a ``func`` that takes this rank's whole share as ONE list and returns one float per item
        (``inference.nELBO_batch``: the share is evaluated side by side on this rank's GPU).  Every rank returns the
        full list, in order."""
        items = list(items)
        n = len(items)
        mine = list(range(self.rank, n, self.world))
        vals = [float(v) for v in func([items[i] for i in mine])] if mine else []
        if len(vals) != len(mine):
            raise ValueError('map_lists: %d results for %d items' % (len(vals), len(mine)))
        if self.world == 1:
            return vals
        import numpy as np
        buf = np.zeros(n)
        buf[mine] = vals
        return [float(v) for v in self._ctx.allreduce_sum(buf)]

    def take_from_highest(self, key, arrays):
        """Every rank offers ``arrays`` (same shapes on every rank) under a priority ``key`` (distinct over the ranks
        that offer; < 0: nothing to offer, the arrays' CONTENT is then ignored).  All ranks return the arrays of the
        rank with the highest key, or None when no rank offers: two collectives, whatever the outcome."""
        import numpy as np
        top = self._ctx.barrier_max(float(key)) if self.world > 1 else float(key)
        out = []
        for a in arrays:
            a = np.asarray(a, dtype=float)
            buf = a.ravel() if (key >= 0 and float(key) == top) else np.zeros(a.size)
            got = self._ctx.allreduce_sum(buf) if self.world > 1 else np.array(buf)
            out.append(got.reshape(a.shape))
        return out if top >= 0 else None

    def close(self):
        self.comm.cleanup()
