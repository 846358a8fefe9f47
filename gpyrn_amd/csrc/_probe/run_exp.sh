for rep in 1 2; do
for lib in "" "GPRN_HIP_LIB=$PWD/gpyrn_amd/csrc/_probe/libgprn_hip_noprio.so"; do
for cfg in 3 2; do
env $lib python bench.py --no-cpu --no-calc --blocks 5 --config $cfg > gpurun_out/r3_b19.json 2>gpurun_out/r3_b19.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b19.json').read().strip().splitlines()[-1]); print('[${lib:0:12}] cfg $cfg:', round(d['value'],2))"
done; done; done
timeout -k 10 300 python gpyrn_amd/csrc/_probe/probe_grad.py 2>&1 | grep -v "^ELBO=" | tail -40
