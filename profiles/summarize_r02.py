#!/usr/bin/env python3
"""Summaries of the round-2 rocprofv3 runs over bench.py (config 3, one MI355X).

Every launch family of the tile kernel is its own instantiation now (last template argument of
`k_tile_gemm<BM, BN, waves, TRI, TAG>`: 0 panel products, 1 in-panel K=128 updates, 2 next-panel K=512 updates,
3 bulk K=512 updates, 4 X^T X / prediction / diagnostics, 5 the look-ahead part of the bulk updates), so a kernel
trace isolates the bulk launches by name.

    # on the GPU box (rocprofv3: cd /tmp && export TMPDIR=/tmp first)
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof -o runc -- python3 bench.py --no-cpu
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_fetch -o runc -- python3 bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_write -o runc -- python3 bench.py ... (same)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace \\
              --output-format csv -d gpurun_out/r2_pmc_mfma -o runc -- python3 bench.py ... (same)
    python3 profiles/summarize_r02.py families gpurun_out/r2_prof            > profiles/r02_kernel_families.json
    python3 profiles/summarize_r02.py traffic  gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write > profiles/r02_pmc_bulk_update.json
    python3 profiles/summarize_r02.py mfma     gpurun_out/r2_pmc_mfma        > profiles/r02_pmc_mfma_util.json

Counter collection serialises kernels: the library sees ROCPROF_COUNTER_COLLECTION=1 and runs the event schedule
(same kernels, no in-kernel waits).  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request of a wide coalesced read and is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (= 64 x the number of fp64 MFMAs per SIMD stream);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import csv
import glob
import json
import re
import sys

FAMILY = {'0': 'panel products (K=128, triangular X_kk)', '1': 'in-panel updates (K=128)',
          '2': 'next-panel updates (K=512)', '3': 'bulk updates (K=512)', '4': 'X^T X, prediction, diagnostics',
          '5': 'bulk updates, look-ahead part (K=512)'}
TILE = re.compile(r'k_tile_gemm<(\d+), (\d+), (\d+), (\d+), (\d+)>')
N_SIMD = 256 * 4


def _one(d, pattern):
    hits = glob.glob(d + '/**/' + pattern, recursive=True)
    if not hits:
        sys.exit(f'no {pattern} under {d}')
    return hits[0]


def families(d):
    rows = list(csv.DictReader(open(_one(d, '*kernel_trace.csv'))))
    agg = {}
    total = 0.0
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        m = TILE.search(name)
        key = name if not m else 'k_tile_gemm<%sx%s, %s waves%s> : %s' % (
            m.group(1), m.group(2), m.group(3), {'0': '', '1': ', B tri', '2': ', A tri'}[m.group(4)], FAMILY[m.group(5)])
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
        total += dur
    out = [{'kernel': k, 'calls': v[0], 'total_us': round(v[1], 1), 'avg_us': round(v[1] / v[0], 2),
            'min_us': round(v[2], 2), 'max_us': round(v[3], 2), 'share_of_kernel_time': round(v[1] / total, 4)}
           for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    print(json.dumps({'source': 'rocprofv3 --kernel-trace of `python3 bench.py --no-cpu` (config 3, one MI355X)',
                      'kernels': out}, indent=1))


def union(d):
    """Time during which at least one K = 512 launch (next-panel, look-ahead and bulk updates: tags 2, 5, 3) was open,
    per sweep, and the same for the bulk + look-ahead launches alone: the per-launch rate of such a launch is flops over
    the time the launch is OPEN, during which it shares the CUs with up to three other streams' kernels; the union-time
    rate says what the K = 512 work got done at while any of it was running."""
    rows = list(csv.DictReader(open(_one(d, '*kernel_trace.csv'))))
    sweeps = sum(1 for r in rows if r['Kernel_Name'].startswith('k_elbo_final'))

    def cover(tags):
        iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows
                    if (lambda m: m and m.group(5) in tags)(TILE.search(r['Kernel_Name'])))
        tot, cur_a, cur_b = 0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    tot += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            tot += cur_b - cur_a
        return tot / 1e3, len(iv), sum(b - a for a, b in iv) / 1e3
    k512 = cover({'2', '3', '5'})
    bulk = cover({'3', '5'})
    print(json.dumps({'source': 'rocprofv3 --kernel-trace of `python3 bench.py --no-cpu` (config 3, one MI355X)', 'sweeps': sweeps,
                      'k512_launches': k512[1], 'k512_union_us_per_sweep': k512[0] / sweeps, 'k512_open_us_per_sweep': k512[2] / sweeps,
                      'bulk_ahead_launches': bulk[1], 'bulk_ahead_union_us_per_sweep': bulk[0] / sweeps,
                      'bulk_ahead_open_us_per_sweep': bulk[2] / sweeps}, indent=1))


def _bulk(d, counter):
    vals = []
    for r in csv.DictReader(open(_one(d, '*counter_collection.csv'))):
        m = TILE.search(r['Kernel_Name'])
        if r['Counter_Name'] == counter and m and m.group(5) == '3':
            vals.append((float(r['Counter_Value']), int(r['Grid_Size']) // int(r['Workgroup_Size']),
                         (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                         'k_tile_gemm<%s, %s, %s, %s, 3>' % m.groups()[:4]))
    return vals


def traffic(df, dw):
    f, w = _bulk(df, 'FETCH_SIZE'), _bulk(dw, 'WRITE_SIZE')
    f_kib, w_kib = sum(x[0] for x in f) / len(f), sum(x[0] for x in w) / len(w)
    wgs = sum(x[1] for x in f) / len(f)
    print(json.dumps({
        'kernel': f[0][3] + ', bulk-update launches (K=512) of bench.py config 3',
        'launches_fetch_pass': len(f), 'launches_write_pass': len(w), 'avg_workgroups_per_launch': wgs,
        'fetch_size_kib_raw_per_launch': f_kib, 'write_size_kib_per_launch': w_kib,
        'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
        'hbm_bytes_per_launch': (2 * f_kib + w_kib) * 1024,
        'hbm_bytes_per_launch_uncorrected': (f_kib + w_kib) * 1024,
        'avg_launch_us_serialised': sum(x[2] for x in f) / len(f),
    }, indent=1))


def fill(df, dw):
    """HBM traffic of the covariance fill kernels (k_fill_sym<2> SE, <4> QP) from a FETCH_SIZE and a WRITE_SIZE pass."""
    def rows(d, counter):
        out = {}
        for r in csv.DictReader(open(_one(d, '*counter_collection.csv'))):
            if r['Counter_Name'] == counter and 'k_fill_sym<' in r['Kernel_Name']:
                kid = r['Kernel_Name'].split('k_fill_sym<')[1].split('>')[0]
                out.setdefault(kid, []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
        return out
    f, w = rows(df, 'FETCH_SIZE'), rows(dw, 'WRITE_SIZE')
    res = {}
    for kid in sorted(w):
        wk = sum(x[0] for x in w[kid]) / len(w[kid])
        fk = sum(x[0] for x in f.get(kid, [(0.0, 0.0)])) / max(1, len(f.get(kid, [])))
        us = sum(x[1] for x in w[kid]) / len(w[kid])
        res['k_fill_sym<%s>' % kid] = {'launches': len(w[kid]), 'write_size_kib_per_launch': wk, 'fetch_size_kib_raw_per_launch': fk,
                                       'hbm_bytes_per_launch': (2 * fk + wk) * 1024, 'avg_launch_us_under_pmc': us}
    print(json.dumps({'kernels': res, 'algorithmic_bytes_per_launch': '8 N^2 = 134217728 at N = 4096 (one write of the matrix)',
                      'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)'}, indent=1))


def mfma(d):
    names = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_INSTS_VALU_MFMA_F64', 'GRBM_GUI_ACTIVE')
    c = {n: _bulk(d, n) for n in names}
    n = len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    avg = {k: sum(x[0] for x in v) / len(v) for k, v in c.items() if v}
    us = sum(x[2] for x in c['SQ_VALU_MFMA_BUSY_CYCLES']) / n
    clock_ghz = avg['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3)                    # cycles per ns
    simd_cycles = N_SIMD * us * 1e3 * clock_ghz
    print(json.dumps({
        'kernel': c['SQ_VALU_MFMA_BUSY_CYCLES'][0][3] + ', bulk-update launches (K=512), serialised (event schedule)',
        'launches': n, 'avg_launch_us': us, 'counters_avg_per_launch': avg,
        'effective_clock_ghz': clock_ghz,
        'mfma_pipe_busy_fraction': avg['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles,
        'mfma_busy_per_resident_wave_cycle': (avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_WAVE_CYCLES'])
                                              if 'SQ_WAVE_CYCLES' in avg else None),
        'fp64_mfma_per_launch': avg.get('SQ_INSTS_VALU_MFMA_F64'),
        'tflops_from_mfma_count': (avg['SQ_INSTS_VALU_MFMA_F64'] * 2048 / (us * 1e-6) / 1e12
                                   if 'SQ_INSTS_VALU_MFMA_F64' in avg else None),
        'note': 'busy fraction = MFMA busy cycles / (1024 SIMDs x launch time x clock from GRBM_GUI_ACTIVE / 8)',
    }, indent=1))


if __name__ == '__main__':
    what = sys.argv[1]
    {'families': families, 'traffic': traffic, 'mfma': mfma, 'union': union, 'fill': fill}[what](*sys.argv[2:])
