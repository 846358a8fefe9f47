#!/usr/bin/env python3
"""Timeline of the factorisation's latency chain from a rocprofv3 kernel trace (CSV):
for every diagonal-block launch, its duration, and what ran between its end and the start of the next one.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2_tr -o runc -- python3 bench.py --config 2 --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc
    python3 profiles/chain_timeline.py gpurun_out/r2_tr [first_step last_step]
"""
import csv, glob, sys, re

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
diag = [i for i, r in enumerate(rows) if 'k_diag_block' in r[2]]
print(f'{len(rows)} kernels, {len(diag)} diagonal blocks')
# the longest run of diagonal blocks that follow each other within 300 us = one factorisation
gaps = []
for a, b in zip(diag[:-1], diag[1:]):
    s0, e0, _ = rows[a]
    s1, e1, _ = rows[b]
    if s1 - e0 > 300e3:
        continue
    between = [(r[0], r[1], r[2]) for r in rows[a + 1:b]]
    gaps.append((e0 - s0, s1 - e0, s1 - s0, between, a))
import statistics as st
print('diag us: median %.1f  | diag-end -> next diag-start us: median %.1f | step us: median %.1f (n = %d)' % (
    st.median(g[0] for g in gaps) / 1e3, st.median(g[1] for g in gaps) / 1e3, st.median(g[2] for g in gaps) / 1e3, len(gaps)))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 40
hi = int(sys.argv[3]) if len(sys.argv) > 3 else lo + 6
for dd, gap, step, between, idx in gaps[lo:hi]:
    t0 = rows[idx][0]
    print('--- diag %.1f us, then (times relative to the diag start, us):' % (dd / 1e3))
    for s, e, n in between:
        m = re.search(r'k_tile_gemm<(\d+), (\d+), (\d+), (\d+), (\d+)>', n)
        short = ('tile %sx%s w%s tri%s tag%s' % m.groups()) if m else n[:40]
        print('    %-34s start %7.1f  end %7.1f  (%.1f)' % (short, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))

# ---- per factorisation (a run of diagonal blocks): how long, how many steps, how the step times are distributed
runs, cur = [], [diag[0]]
for a, b in zip(diag[:-1], diag[1:]):
    if rows[b][0] - rows[a][1] > 300e3:
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
print('\nfactorisations (runs of diagonal blocks):')
for r in runs[-8:]:
    t0, t1 = rows[r[0]][0], rows[r[-1]][1]
    steps = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(r[:-1], r[1:])]
    if not steps:
        continue
    # batch = diagonal-block launches have one workgroup per matrix; not in the trace: report durations only
    q = sorted(steps)
    print('  %3d steps, %7.1f us first-diag-start -> last-diag-end; step us: min %.1f  median %.1f  p90 %.1f  max %.1f' % (
        len(r), (t1 - t0) / 1e3, q[0], q[len(q) // 2], q[int(len(q) * 0.9)], q[-1]))

# ---- the chain of the longest run, step by step: diag | wait | L | wait | U | wait (us)
best = max(runs, key=len)
print('\nlongest run (%d diagonal blocks), per step: diag | ->L | L | ->U | U | ->next diag   [other kernels running at the diag start]' % len(best))
def find(after, before, pat):
    for r in rows:
        if r[0] >= after and r[0] < before and pat in r[2]:
            return r
    return None
for a, b in zip(best[:-1], best[1:]):
    s0, e0, _ = rows[a]
    s1 = rows[b][0]
    L = find(s0, s1, 'k_chain_l')
    U = find(s0, s1, 'k_chain_u')
    busy = sum(1 for r in rows if r[0] < s0 < r[1] and 'k_diag' not in r[2])
    if L and U:
        print('  %6.1f | %5.1f | %5.1f | %5.1f | %5.1f | %6.1f   = %6.1f  [%d]' % (
            (e0 - s0) / 1e3, (L[0] - e0) / 1e3, (L[1] - L[0]) / 1e3, (U[0] - L[1]) / 1e3, (U[1] - U[0]) / 1e3,
            (s1 - U[1]) / 1e3, (s1 - s0) / 1e3, busy))
    else:
        print('  %6.1f | (no chain products in this step)  = %6.1f  [%d]' % ((e0 - s0) / 1e3, (s1 - s0) / 1e3, busy))

# ---- what runs while the chain stands still after a step's update (gaps > 30 us in the longest run)
print('\nkernels running between the end of a step\'s update and the start of the next diagonal block (gaps > 30 us):')
shown = 0
for a, b in zip(best[:-1], best[1:]):
    s0 = rows[a][0]; s1 = rows[b][0]
    U = find(s0, s1, 'k_chain_u')
    if not U or s1 - U[1] < 30e3 or shown >= 3:
        continue
    shown += 1
    print('  gap %.1f us after the step that started at %.1f us of the run:' % ((s1 - U[1]) / 1e3, (s0 - rows[best[0]][0]) / 1e3))
    agg = {}
    for r in rows:
        if r[1] > U[1] and r[0] < s1:
            m = TILE_RE.search(r[2]) if 'TILE_RE' in globals() else None
            key = r[2][:60]
            e = agg.setdefault(key, [0, 0.0, 1e18, 0])
            e[0] += 1; e[1] += (min(r[1], s1) - max(r[0], U[1])) / 1e3; e[2] = min(e[2], r[0]); e[3] = max(e[3], r[1])
    for k, e in sorted(agg.items(), key=lambda kv: kv[1][2]):
        print('      %-62s x%d  first start %+7.1f  last end %+7.1f (us rel. to the update\'s end)' % (k, e[0], (e[2] - U[1]) / 1e3, (e[3] - U[1]) / 1e3))
