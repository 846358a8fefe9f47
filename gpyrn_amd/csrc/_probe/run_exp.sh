python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_parity_gpu.py -x -q -k "forced_sweeps or trajectory_at_baseline or schedule_and" 2>&1 | tail -3
for v in "" "GPRN_SPLIT_INNER=0" "GPRN_LOWER_DIAG=0" "GPRN_SPLIT_INNER=0 GPRN_LOWER_DIAG=0" "GPRN_SPLIT_INNER=2"; do
env $v python bench.py --no-cpu --no-calc --blocks 3 > gpurun_out/r3_b17.json 2>gpurun_out/r3_b17.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b17.json').read().strip().splitlines()[-1]); print('[$v] cfg3:', round(d['value'],2), d['elbo_last'])"
done
for v in "" "GPRN_SPLIT_INNER=0 GPRN_LOWER_DIAG=0"; do
env $v python bench.py --no-cpu --no-calc --blocks 3 --config 2 > gpurun_out/r3_b17.json 2>gpurun_out/r3_b17.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b17.json').read().strip().splitlines()[-1]); print('[$v] cfg2:', round(d['value'],2))"
env $v python bench.py --no-cpu --no-calc --blocks 3 --config 4 > gpurun_out/r3_b17.json 2>gpurun_out/r3_b17.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b17.json').read().strip().splitlines()[-1]); print('[$v] cfg4:', round(d['value'],2))"
done
