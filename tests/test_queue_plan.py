"""The task graph of the dataflow schedule (gpyrn_amd/csrc/queue.hip) checked on the host: no GPU.

The graph is derived from the sequential blocked algorithm (Cholesky + inverse factor, csrc/factor.hip; it replaces
the reference's LU solve / Cholesky / cho_solve of meanfield.py:771,850,1087-1090,1041,1051).  Here every node is
executed with NumPy on a real matrix in orders the device is free to choose -- random topological orders and the two
extreme ones -- and the result must be the Cholesky factor and its inverse each time: a missing edge (read-after-write,
write order, write-after-read) shows as a wrong factor for some order.
"""
import numpy as np
import pytest

from gpyrn_amd import _hip

TILE = 128
BUF_B, BUF_X = 0, 1


def _view(buf, off, ld, rows, cols):
    r0, c0 = divmod(int(off), ld)
    return buf[r0:r0 + rows, c0:c0 + cols]


def _run(op, bufs, ld):
    kind, _cls, _nent, flags, c_buf, a_buf, b_buf, modes, c_off, a_off, b_off, klen = (int(v) for v in op)
    B, X = bufs[BUF_B], bufs[BUF_X]
    if kind == 3:                                        # the chain's kernels of tile step k = klen
        k = klen
        s = slice(k * TILE, (k + 1) * TILE)
        if c_buf == 0:                                   # diag: L_kk = chol(B_kk) (lower part of the tile), X_kk = L_kk^-1
            sym = np.tril(B[s, s]) + np.tril(B[s, s], -1).T
            L = np.linalg.cholesky(sym)
            B[s, s] = L + np.triu(B[s, s], 1)
            X[s, s] = np.tril(np.linalg.inv(L))
        elif c_buf == 1:                                 # L_{k+1,k} = B_{k+1,k} X_kk^T, in place
            n = slice((k + 1) * TILE, (k + 2) * TILE)
            B[n, s] = B[n, s] @ X[s, s].T
        else:                                            # B_{k+1,k+1} -= L_{k+1,k} L_{k+1,k}^T (the lower part is what is read)
            n = slice((k + 1) * TILE, (k + 2) * TILE)
            upd = B[n, s] @ B[n, s].T
            B[n, n] -= np.tril(upd)
        return
    c_mode, a_mode, b_mode = modes & 3, (modes >> 2) & 1, (modes >> 3) & 1
    A = _view(bufs[a_buf], a_off, ld, TILE, klen) if a_mode == 0 else _view(bufs[a_buf], a_off, ld, klen, TILE).T
    Bm = _view(bufs[b_buf], b_off, ld, TILE, klen).T if b_mode == 0 else _view(bufs[b_buf], b_off, ld, klen, TILE)
    C = _view(bufs[c_buf], c_off, ld, TILE, TILE)
    P = A @ Bm
    new = {0: P, 1: C - P, 2: -P}[c_mode]
    if flags & 1:                                        # diagonal SYRK tile: the upper-right quarter is not computed
        new[:64, 64:] = C[:64, 64:]
    C[...] = new


def _spd(n, seed):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.uniform(0, 0.4 * n, n))
    r = t[:, None] - t[None, :]
    return np.exp(-0.5 * r**2 / 20.0**2) * np.outer(1 + 0.1 * rng.rand(n), np.ones(n)) ** 0 + (1.0 + rng.rand(n)) * np.eye(n)


def _orders(n, preds_count, succ, how, rng):
    left = preds_count.copy()
    ready = [v for v in range(n) if left[v] == 0]
    while ready:
        if how == 'random':
            i = rng.randint(len(ready))
        elif how == 'last':
            i = int(np.argmax(ready))
        else:
            i = int(np.argmin(ready))
        v = ready.pop(i)
        yield v
        for s in succ[v]:
            left[s] -= 1
            if left[s] == 0:
                ready.append(s)


@pytest.mark.parametrize('T,outer', [(1, 4), (2, 4), (5, 2), (6, 4), (7, 3), (9, 4), (6, 16)])
def test_any_order_the_graph_allows_factors_the_matrix(T, outer):
    ops, edges = _hip.queue_plan(T, outer)
    n = ops.shape[0]
    assert np.all(edges[:, 0] < edges[:, 1])             # program order is a topological order: no cycles
    assert len({(int(a), int(b)) for a, b in edges}) == len(edges)
    succ = [[] for _ in range(n)]
    indeg = np.zeros(n, dtype=int)
    for a, b in edges:
        succ[int(a)].append(int(b))
        indeg[int(b)] += 1
    assert indeg.max() < 2**16 and np.bincount(edges[:, 0], minlength=n).max() <= 1024
    # exactly one node without inputs: the first diagonal block
    assert list(np.nonzero(indeg == 0)[0]) == [0] and ops[0, 0] == 3 and ops[0, 4] == 0
    N = ld = T * TILE
    A = _spd(N, 100 + T)
    Lref = np.linalg.cholesky(A)
    Xref = np.linalg.inv(Lref)
    rng = np.random.RandomState(7 * T + outer)
    for how in ['first', 'last'] + ['random'] * 4:
        bufs = {BUF_B: A.copy(), BUF_X: np.zeros((N, N))}
        done = 0
        for v in _orders(n, indeg, succ, how, rng):
            _run(ops[v], bufs, ld)
            done += 1
        assert done == n                                  # every node became ready: nothing waits for ever
        np.testing.assert_allclose(np.tril(bufs[BUF_B]), Lref, rtol=0, atol=1e-11, err_msg=how)
        np.testing.assert_allclose(np.tril(bufs[BUF_X]), Xref, rtol=0, atol=1e-10, err_msg=how)


def test_graph_of_the_headline_configuration():
    """N = 4096 (T = 32, outer panels of 4 tiles): sizes the device code relies on (successors of one node in one
    wave-wide pass or a few, 16-bit dependency counters, 21-bit node ids) and the flop count of the tile nodes."""
    T = 32
    ops, edges = _hip.queue_plan(T, 4)
    n = ops.shape[0]
    assert n < 2**21
    outdeg = np.bincount(edges[:, 0], minlength=n)
    indeg = np.bincount(edges[:, 1], minlength=n)
    assert outdeg.max() <= 128 and indeg.max() <= 16
    chain = ops[ops[:, 0] == 3]
    assert len(chain) == T + 2 * (T - 1)
    # classes: 0 = feeds the chain's next steps ... 3 = far trailing update; every class is used
    assert set(np.unique(ops[ops[:, 0] != 3, 1])) == {0, 1, 2, 3}
    # tile-node flops = 2/3 N^3 minus the chain's own share (diag, L_{k+1,k}, B_{k+1,k+1}: O(T) tiles), with the
    # diagonal SYRK tiles at 3/4 and the panel products' zero halves counted as the kernels compute them
    tile = ops[ops[:, 0] == 0]
    fl = 2.0 * TILE * TILE * tile[:, 11] * np.where(tile[:, 3] & 1, 0.75, 1.0)
    panel = ops[(ops[:, 0] == 1) | (ops[:, 0] == 2)]
    fl_panel = 2.0 * TILE**3 * len(panel)
    N = T * TILE
    # (what the kernels execute is a few per cent above the algorithmic 2/3 N^3: whole tiles on and next to the diagonal)
    assert 1.0 * (2 / 3) * N**3 < fl.sum() + fl_panel < 1.09 * (2 / 3) * N**3
