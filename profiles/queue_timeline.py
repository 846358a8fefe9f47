#!/usr/bin/env python3
"""Reduce a trace of the dataflow schedule (GPRN_QUEUE_TRACE=n; csrc/queue.hip) to a timeline per factorisation:
per tile step when the chain's three kernels waited / ran, how long ready entries of each class waited for a worker,
how long they ran, how busy the workers were.

    GPRN_QUEUE_TRACE=400000 GPRN_QUEUE_TRACE_FILE=gpurun_out/queue_trace.bin python profiles/queue_trace_run.py
    python profiles/queue_timeline.py gpurun_out/queue_trace.bin
"""
import sys

import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint64)
    magic, n, nops, T = (int(v) for v in raw[:4])
    assert magic == 0x47505251
    rec = raw[4:4 + 4 * n].reshape(n, 4)
    ops = raw[4 + 4 * n:].view(np.int32).reshape(-1, 4)[:nops]
    return rec, ops, T


def main(path):
    rec, ops, T = load(path)
    who = rec[:, 0]
    call = (who >> np.uint64(56)).astype(int)
    worker = ((who >> np.uint64(32)) & np.uint64(0xffff)).astype(int)
    entry = (who & np.uint64(0xffffffff)).astype(np.int64)
    m = (entry >> 24) & 0xff
    sub = (entry >> 21) & 7
    op = entry & 0x1fffff
    t1, t2, t3 = (rec[:, i].astype(np.int64) for i in (1, 2, 3))
    print(f'{len(rec)} records, {len(ops)} nodes, T = {T}, calls {sorted(set(call))}')
    for cid in sorted(set(call)):
        sel = call == cid
        is_chain = sel & (worker == 0xffff)
        is_tile = sel & (worker != 0xffff)
        if not is_tile.any():
            continue
        tmin = min(t1[sel].min(), t2[sel].min())
        us = lambda t: (t - tmin) * 0.01
        nb = int(m[sel].max()) + 1
        span = us(t3[sel].max())
        print(f'\n== call {cid}: {nb} matrices, {is_tile.sum()} entries on {len(set(worker[is_tile]))} workers, {span:.0f} us')
        # workers: busy share
        busy = (t3[is_tile] - t1[is_tile]).sum() * 0.01
        run = (t3[is_tile] - t2[is_tile]).sum() * 0.01
        nw = len(set(worker[is_tile]))
        print(f'   workers: claim..done {busy / (nw * span):.1%} of worker-time, start..done {run / (nw * span):.1%}')
        for cls in range(5):
            k = is_tile & (ops[np.minimum(op, len(ops) - 1), 1] == cls)
            if not k.any():
                continue
            d_run = (t3[k] - t2[k]) * 0.01
            d_pre = (t2[k] - t1[k]) * 0.01
            print(f'   class {cls}: {k.sum():6d} entries, found->start {np.median(d_pre):5.1f} us, start->done median {np.median(d_run):6.1f} '
                  f'p90 {np.percentile(d_run, 90):6.1f} us')
        # chain per step (matrix 0)
        ch = is_chain & (m == 0)
        typ = ops[np.minimum(op, len(ops) - 1), 3]
        step = ops[np.minimum(op, len(ops) - 1), 2]
        rows = []
        for k in range(T):
            row = [k]
            for ty in (0, 1, 2):
                s = ch & (typ == ty) & (step == k) & (ops[np.minimum(op, len(ops) - 1), 0] == 3)
                if s.any():
                    i = np.nonzero(s)[0][0]
                    row += [us(t1[i]), (t2[i] - t1[i]) * 0.01, (t3[i] - t2[i]) * 0.01]
                else:
                    row += [np.nan] * 3
            rows.append(row)
        rows = np.array(rows)
        print('   step | diag: start  wait   run | L: start  wait   run | U: start  wait   run | step time')
        prev = None
        for r in rows:
            dt = r[1] - prev if prev is not None else np.nan
            prev = r[1]
            print(f'   {int(r[0]):4d} | {r[1]:9.1f} {r[2]:6.1f} {r[3]:5.1f} | {r[4]:8.1f} {r[5]:6.1f} {r[6]:5.1f} | {r[7]:8.1f} {r[8]:6.1f} {r[9]:5.1f} | {dt:6.1f}')
        steps = np.diff(rows[:, 1])
        print(f'   step time: median {np.nanmedian(steps):.1f} us, mean {np.nanmean(steps):.1f}; waits per step: diag {np.nanmean(rows[:, 2]):.1f} '
              f'L {np.nanmean(rows[:, 5]):.1f} U {np.nanmean(rows[:, 8]):.1f} us')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/queue_trace.bin')
