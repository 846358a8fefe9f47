#!/usr/bin/env python3
"""Where a sweep's time goes, phase by phase, from a rocprofv3 kernel trace (CSV).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o runc -- python3 bench.py --no-cpu --no-calc --steps 6 --warmup 2 --blocks 1
    python3 profiles/phase_timeline.py DIR [sweep_index] [--kernels]

A sweep = k_prep_nodes ... the next k_prep_nodes (the last sweep of a call: ... its k_elbo_final).  Per phase (node / weight half-sweep): head (prep start -> first diagonal
block starts), factorisation (first diagonal block -> last chain kernel ends), tail (-> the next phase's prep /
k_elbo_final ends); with --kernels every kernel of head and tail with its start / end relative to the phase start.
"""
import csv
import glob
import statistics as st
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else -2
show = '--kernels' in sys.argv
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()


def idx(name):
    return [i for i, r in enumerate(rows) if r[2].startswith(name)]


pn, pw, ef = idx('k_prep_nodes'), idx('k_prep_weights'), idx('k_elbo_final')
sweeps = []
for n, a in enumerate(pn):
    # (a sweep's ELBO assembly may run beside the NEXT sweep's node phase -- option "overlap", bit 16 -- so the weight
    # phase is taken to end where the next sweep's preparation starts; the last sweep of a call ends with its k_elbo_final)
    w = [i for i in pw if i > a]
    nxt = pn[n + 1] if n + 1 < len(pn) else None
    if not w or (nxt is not None and w[0] > nxt):
        continue
    if nxt is None or rows[nxt][0] - rows[w[0]][0] > 50e6:          # the next call's sweep: close with this call's last k_elbo_final
        e = [i for i in ef if i > w[0] and (nxt is None or i < nxt)]
        if not e:
            continue
        sweeps.append((a, w[0], e[-1]))
    else:
        sweeps.append((a, w[0], nxt - 1))
print('%d kernels, %d sweeps' % (len(rows), len(sweeps)))
CHAIN = ('k_diag_block', 'k_chain_l', 'k_chain_u', 'k_chain_step')


def phase(a, b, label):
    """kernels [a, b): a = the phase's prep, b = the next phase's prep (or one past k_elbo_final)"""
    t0 = rows[a][0]
    ks = rows[a:b]
    chain = [k for k in ks if k[2].startswith(CHAIN)]
    if not chain:
        return None
    f0 = min(k[0] for k in chain if k[2].startswith('k_diag_block'))
    f1 = max(k[1] for k in chain)
    end = rows[b][0] if b < len(rows) else max(k[1] for k in ks)
    return {'label': label, 'head': (f0 - t0) / 1e3, 'factor': (f1 - f0) / 1e3, 'tail': (end - f1) / 1e3, 'total': (end - t0) / 1e3,
            't0': t0, 'f0': f0, 'f1': f1, 'end': end, 'ks': ks}


stats = {'node': [], 'weight': []}
for a, w, e in sweeps:
    for ph in (phase(a, w, 'node'), phase(w, e + 1, 'weight')):
        if ph:
            stats[ph['label']].append(ph)
for lab in ('node', 'weight'):
    v = stats[lab]
    if v:
        print('%-6s phase, median over %d: head %.1f us | factorisation %.1f us | tail %.1f us | total %.1f us' % (
            lab, len(v), st.median(x['head'] for x in v), st.median(x['factor'] for x in v), st.median(x['tail'] for x in v),
            st.median(x['total'] for x in v)))
if sweeps:
    print('sweep (prep_nodes -> next prep_nodes), median: %.1f us' % st.median(
        (rows[b[0]][0] - rows[a[0]][0]) / 1e3 for a, b in zip(sweeps[:-1], sweeps[1:])) if len(sweeps) > 1 else '')
if show and sweeps:
    a, w, e = sweeps[which]
    for ph in (phase(a, w, 'node'), phase(w, e + 1, 'weight')):
        if not ph:
            continue
        print('\n== %s phase of sweep %d: head %.1f, factorisation %.1f, tail %.1f us' % (ph['label'], which, ph['head'], ph['factor'], ph['tail']))
        for s, en, n in ph['ks']:
            if en <= ph['f0'] + 150e3 or s >= ph['f1'] - 150e3:
                print('   %-60s start %8.1f  end %8.1f  (%.1f)' % (n[:60], (s - ph['t0']) / 1e3, (en - ph['t0']) / 1e3, (en - s) / 1e3))
