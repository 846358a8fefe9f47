#!/bin/bash
# Everything under profiles/r03_* on the GPU box, in two calls (a gpurun call is limited to 20 minutes):
#   bash profiles/run_r03_profiles.sh a     the bench line, kernel trace + statistics, PMC passes
#   bash profiles/run_r03_profiles.sh b     the other configurations, per-rank shapes, the opt-in schedules
# (results land in gpurun_out/final/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-a}
if [ "$PART" = "a" ]; then
# --- the bench line (CPU baseline: three sweeps on all host cores, then the GPU blocks)
timeout -k 10 1100 python bench.py > $O/r03_bench.json 2> $O/bench.err
# --- kernel trace + stats of the same command without the CPU leg; per-family table; chain timeline; union time of the K = 512 launches
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --blocks 5 > $GRAFT_REPO_ROOT/$O/r03_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py families $O/prof > $O/r03_kernel_families.json
python3 profiles/summarize_r02.py union $O/prof > $O/r03_k512_union.json
python3 profiles/chain_timeline.py $O/prof 300 302 > $O/r03_chain_timeline_cfg3.txt
python3 profiles/phase_timeline.py $O/prof -2 --kernels > $O/r03_phase_timeline_cfg3.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
rm -rf $O/prof
# --- PMC passes (separate runs; counter collection serialises kernels -> the library uses the event schedule)
for pmc in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$pmc -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_$pmc.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
done
python3 profiles/summarize_r02.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/r03_pmc_bulk_update.json
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmc_mfma.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
python3 profiles/summarize_r02.py mfma $O/pmc_mfma > $O/r03_pmc_mfma_util.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
# --- traffic of the covariance fill kernels (one set-up: 6 SE + 2 QP fills)
for pmc in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmcf_$pmc -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --blocks 1 --no-cpu --no-calc > $GRAFT_REPO_ROOT/$O/pmcf_$pmc.json 2>> $GRAFT_REPO_ROOT/$O/prof.err )
done
python3 profiles/summarize_r02.py fill $O/pmcf_FETCH_SIZE $O/pmcf_WRITE_SIZE > $O/r03_pmc_fill.json
rm -rf $O/pmcf_FETCH_SIZE $O/pmcf_WRITE_SIZE
fi
if [ "$PART" = "b" ]; then
# --- the other configs on one GPU, the two-rank rehearsal
timeout -k 10 300 python bench.py --no-cpu --config 2 > $O/r03_bench_cfg2.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --no-cpu --no-calc --config 4 --blocks 5 > $O/r03_bench_cfg4_one_gpu.json 2>> $O/bench.err
timeout -k 10 1000 python bench.py --no-cpu --no-calc --config 5 --steps 5 --warmup 1 --blocks 3 > $O/r03_bench_cfg5_one_gpu.json 2>> $O/bench.err
GPRN_COMM_TRANSPORT=shm timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu --no-calc --blocks 5 > $O/r03_bench_selflaunch_2ranks_shm.json 2>> $O/bench.err
# --- the dataflow schedule (opt-in): the bench line, worker statistics, a timeline of two sweeps, the contraction rate
GPRN_QUEUE=1 GPRN_QUEUE_STATS=1 timeout -k 10 300 python bench.py --no-cpu --no-calc --blocks 5 > $O/r03_bench_dataflow_schedule.json 2> $O/r03_dataflow_worker_stats.txt
GPRN_QUEUE=1 GPRN_QUEUE_TRACE=600000 GPRN_QUEUE_TRACE_FILE=$O/qt_c3.bin timeout -k 10 100 python profiles/queue_trace_run.py 3 > /dev/null 2>&1
python3 profiles/queue_timeline.py $O/qt_c3.bin > $O/r03_dataflow_timeline_cfg3.txt 2>&1
GPRN_QUEUE=1 GPRN_QUEUE_TRACE=600000 GPRN_QUEUE_TRACE_FILE=$O/qt_c2.bin timeout -k 10 100 python profiles/queue_trace_run.py 2 > /dev/null 2>&1
python3 profiles/queue_timeline.py $O/qt_c2.bin > $O/r03_dataflow_timeline_cfg2.txt 2>&1
rm -f $O/qt_c3.bin $O/qt_c2.bin
GPRN_QUEUE_STATS=1 timeout -k 10 200 python gpyrn_amd/csrc/_probe/probe_qrate.py > $O/r03_contraction_rate_launch_vs_worker.txt 2>&1
# --- what one rank of 2 / 4 / 8 sees of config 3 (1 node + 3 / 2 / 1 weights), the sequential head / tail, the block schedule
for sh in 4096,3,1 4096,2,1 4096,1,1; do
  timeout -k 10 200 python bench.py --no-cpu --no-calc --blocks 5 --shape $sh > $O/r03_bench_shape_${sh//,/_}.json 2>> $O/bench.err
done
GPRN_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu --no-calc --blocks 5 > $O/r03_bench_overlap_off.json 2>> $O/bench.err
GPRN_BLOCK_SCHED=1 timeout -k 10 200 python bench.py --no-cpu --no-calc --blocks 5 > $O/r03_bench_block_schedule.json 2>> $O/bench.err
GPRN_BLOCK_SCHED=1 GPRN_BLOCK_MIN_BATCH=3 timeout -k 10 200 python bench.py --no-cpu --no-calc --blocks 5 > $O/r03_bench_block_schedule_weight_phase_only.json 2>> $O/bench.err
fi
ls -la $O
