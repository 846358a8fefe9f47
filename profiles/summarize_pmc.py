#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes over `bench.py --no-cpu` into the HBM traffic of
the dominant kernel (the bulk-update launches of the sweeps: k_tile_gemm<64,64> with the grid sizes of the
"rest" task lists at T = 32 tiles, outer panels of 4, batches of 2 and 6 matrices).

    export GPRN_FLAGS=0     # counter collection serialises kernels: the device-side dependency waits of the
                            # default schedule would time out; the event-based schedule runs the same kernels
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-calc
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-calc
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_pmc_bulk_update.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read, so it is doubled (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is exact.
"""
import csv
import glob
import json
import sys


def bulk_grid_sizes(T=32, outer=4, batches=(2, 6)):
    """threads per bulk launch: 4 workgroups of 256 threads per task of the "rest" class (csrc/factor.hip)"""
    sizes = set()
    for k0 in range(0, T, outer):
        k1 = min(T, k0 + outer)
        n1 = min(T, k1 + outer)
        nrest = sum((i - n1 + 1) + k1 for i in range(n1, T))
        for b in batches:
            if nrest:
                sizes.add(nrest * 4 * 256 * b)
    return sizes


BULK = bulk_grid_sizes()


def launches(d, name):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name and 'k_tile_gemm<64, 64' in r['Kernel_Name'] and int(r['Grid_Size']) in BULK:
            out.append((int(r['Dispatch_Id']), int(r['Grid_Size']) // 256, float(r['Counter_Value'])))
    return out


fetch = launches(sys.argv[1], 'FETCH_SIZE')
write = launches(sys.argv[2], 'WRITE_SIZE')
nf, nw = len(fetch), len(write)
f_kib = sum(x[2] for x in fetch) / nf
w_kib = sum(x[2] for x in write) / nw
wgs = sum(x[1] for x in fetch) / nf
print(json.dumps({
    'kernel': 'k_tile_gemm<64,64>, bulk-update launches (K=512) of bench.py config 3',
    'launches_fetch_pass': nf, 'launches_write_pass': nw, 'avg_workgroups_per_launch': wgs,
    'fetch_size_kib_raw_per_launch': f_kib, 'write_size_kib_per_launch': w_kib,
    'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
    'hbm_bytes_per_launch': (2 * f_kib + w_kib) * 1024,
    'hbm_bytes_per_launch_uncorrected': (f_kib + w_kib) * 1024,
    'algorithmic_c_tile_bytes_per_launch': wgs * 64 * 64 * 8 * 2,
}, indent=1))
