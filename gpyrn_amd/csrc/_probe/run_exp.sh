export GPRN_QUEUE_DEBUG=1
BUDGET_MS=500 timeout -k 10 100 python gpyrn_amd/csrc/_probe/probe_queue.py 2>&1 | tail -7
timeout -k 10 200 python gpyrn_amd/csrc/_probe/probe_qrate2.py 2>&1 | grep TF
for cfg in 3 2 4; do timeout -k 10 150 python bench.py --no-cpu --no-calc --blocks 3 --config $cfg > gpurun_out/r3_b15_c$cfg.json 2>gpurun_out/r3_b15_c$cfg.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b15_c$cfg.json').read().strip().splitlines()[-1]); print('cfg $cfg:', d['value'], d['elbo_last'])"; head -3 gpurun_out/r3_b15_c$cfg.err; done
GPRN_QUEUE_TRACE=600000 GPRN_QUEUE_TRACE_FILE=gpurun_out/qt_c2.bin timeout -k 10 100 python profiles/queue_trace_run.py 2 > gpurun_out/qt_c2.log 2>&1
GPRN_QUEUE_TRACE=600000 GPRN_QUEUE_TRACE_FILE=gpurun_out/qt_c3.bin timeout -k 10 100 python profiles/queue_trace_run.py 3 > gpurun_out/qt_c3.log 2>&1
