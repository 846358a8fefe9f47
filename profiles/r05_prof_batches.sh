cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests -x -q -m gpu -k "batch or side_by_side or small_path or elbocalc_loop or failed or tile_edges or trajectory" > gpurun_out/r05_t3.log 2>&1; tail -4 gpurun_out/r05_t3.log
cd /tmp
for cfg in "45 1 1 256" "512 3 2 32"; do
  set -- $cfg
  tag=n$1_b$4
  GPRN_BATCH_TIMERS=1 python3 $R/profiles/batch_run.py $cfg 3 > $R/gpurun_out/r05_batch_${tag}_plain.log 2>&1
  GPRN_BATCH_TIMERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_batch_${tag} -o t -- python3 $R/profiles/batch_run.py $cfg 3 > $R/gpurun_out/r05_batch_${tag}_prof.log 2>&1
  tail -3 $R/gpurun_out/r05_batch_${tag}_plain.log
done
cd $R && GPRN_SMALL_STAMPS=1 python bench.py --latency --no-cpu --latency-only 45 > gpurun_out/r05_latency_c.jsonl 2> gpurun_out/r05_latency_c.err; tail -4 gpurun_out/r05_latency_c.err; python -c "
import json
for l in open('gpurun_out/r05_latency_c.jsonl'):
    d = json.loads(l); s = d.get('side_by_side') or {}
    print(d['config']['workload'], '| one by one %.1f /s (%.3f ms)' % (d['value'], d['ms_per_evaluation']), '| side', s.get('evaluations'), '%.0f /s' % s.get('value', 0))
"
