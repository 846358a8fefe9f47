#!/bin/bash
# Everything under profiles/r02_* in one go, on the GPU box:  bash profiles/run_r02_profiles.sh  (results land in gpurun_out/final/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
# the stand-alone probes (binaries are not tracked)
for f in diag_bench base16_bench rows_bench lat_bench; do
  [ -x gpyrn_amd/csrc/_probe/$f ] || ( cd gpyrn_amd/csrc/_probe && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include $f.hip -o $f )
done
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 900 python bench.py > $O/r02_bench.json 2> $O/bench.err
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/$O/r02_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py families $O/prof > $O/r02_kernel_families.json
python3 profiles/chain_timeline.py $O/prof 300 302 > $O/r02_chain_timeline_cfg3.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
rm -rf $O/prof
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --config 2 --steps 5 --warmup 2 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/cfg2_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/chain_timeline.py $O/prof2 60 63 > $O/r02_chain_timeline_cfg2.txt
rm -rf $O/prof2
for pmc in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$pmc -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_$pmc.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
done
python3 profiles/summarize_r02.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/r02_pmc_bulk_update.json
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_mfma.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py mfma $O/pmc_mfma > $O/r02_pmc_mfma_util.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
timeout -k 10 300 python bench.py --no-cpu --config 2 > $O/r02_bench_cfg2.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --no-cpu --no-calc --config 4 > $O/r02_bench_cfg4_one_gpu.json 2>> $O/bench.err
./gpyrn_amd/csrc/_probe/diag_bench > $O/r02_diag_kernel_timeline.txt 2>&1
./gpyrn_amd/csrc/_probe/base16_bench > $O/r02_base16_bench.txt 2>&1
./gpyrn_amd/csrc/_probe/rows_bench > $O/r02_chain_products_timeline.txt 2>&1
./gpyrn_amd/csrc/_probe/lat_bench > $O/r02_instruction_latencies.txt 2>&1
GPRN_COMM_TRANSPORT=shm timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu --no-calc > $O/r02_bench_selflaunch_2ranks_shm.json 2>> $O/bench.err
ls -la $O
