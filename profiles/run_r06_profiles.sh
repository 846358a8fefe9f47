#!/bin/bash
# Everything under profiles/r06_* on the GPU box, in two calls (a gpurun call is limited to 20 minutes):
#   bash profiles/run_r06_profiles.sh a     the bench line, kernel trace + statistics, PMC passes
#   bash profiles/run_r06_profiles.sh b     the other configurations, per-rank shapes, the small-N latency lines
# (results land in gpurun_out/final/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-a}
if [ "$PART" = "a" ]; then
# --- the bench line (CPU baseline: three sweeps on all host cores, then the GPU blocks)
timeout -k 10 1100 python bench.py > $O/r06_bench.json 2> $O/bench.err
# --- kernel trace + stats of the same command without the CPU leg; per-family table; union time of the K = 512 launches
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --blocks 5 > $GRAFT_REPO_ROOT/$O/r06_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py families $O/prof > $O/r06_kernel_families.json
python3 profiles/summarize_r02.py union $O/prof > $O/r06_k512_union.json
python3 profiles/phase_timeline.py $O/prof -2 --kernels > $O/r06_phase_timeline_cfg3.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
rm -rf $O/prof
# --- PMC passes (separate runs; counter collection serialises kernels -> the library uses the event schedule)
for pmc in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$pmc -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_$pmc.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
done
python3 profiles/summarize_r02.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/r06_pmc_bulk_update.json
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_mfma.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py mfma $O/pmc_mfma > $O/r06_pmc_mfma_util.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
fi
if [ "$PART" = "b" ]; then
# --- the other configs on one GPU, the rehearsals over the shared-memory transport
timeout -k 10 300 python bench.py --no-cpu --config 2 > $O/r06_bench_cfg2.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --no-cpu --no-calc --config 4 --blocks 5 > $O/r06_bench_cfg4_one_gpu.json 2>> $O/bench.err
timeout -k 10 1000 python bench.py --no-cpu --no-calc --config 5 --steps 5 --warmup 1 --blocks 3 > $O/r06_bench_cfg5_one_gpu.json 2> $O/r06_bench_cfg5_one_gpu.err
# (4 ranks: config 3 sharded + BASELINE config 4 in its stated topology under configs_at_n_gpus; 2 ranks: config 5's shape at N = 2048)
GPRN_COMM_TRANSPORT=shm timeout -k 10 300 python3 bench.py --gpus 4 --no-cpu --no-calc --blocks 3 --steps 5 > $O/r06_bench_4ranks_shm_one_gpu.json 2>> $O/bench.err
# (six ranks -- the most a one-GPU box's process guard admits: config 5's partition features, see tests/test_sharding.py for eight)
GPRN_COMM_TRANSPORT=shm timeout -k 10 400 python3 bench.py --gpus 6 --no-cpu --no-calc --blocks 3 --steps 5 --also-config 5:2048 > $O/r06_bench_6ranks_shm_one_gpu.json 2>> $O/bench.err
# --- what one rank of 2 / 4 / 8 sees of config 3 (1 node + 3 / 2 / 1 weights)
for sh in 4096,3,1 4096,2,1 4096,1,1; do
  timeout -k 10 200 python bench.py --no-cpu --no-calc --blocks 5 --shape $sh > $O/r06_bench_shape_${sh//,/_}.json 2>> $O/bench.err
done
# --- the small-N regime: nELBO evaluations per second, one by one and side by side, the host baseline beside them
timeout -k 10 500 python bench.py --latency > $O/r06_latency.jsonl 2>> $O/bench.err
GPRN_SMALL_STAMPS=1 timeout -k 10 100 python bench.py --latency --no-cpu --latency-only 45 --latency-reps 20 > /dev/null 2> $O/r06_small_path_stamps.txt
# --- the accuracy / cost of the substitution panels: profiles/factor_accuracy_probe.py, setup_time_probe.py, prior_term_accuracy_diag.py
fi
ls -la $O
