for v in "" "GPRN_BULK_SHAPE_BIG=0" "GPRN_BULK_SHAPE_BIG=0 GPRN_BULK_PAD_KB=0" "GPRN_BULK_SHAPE_BIG=0 GPRN_BULK_PAD_KB=24" "GPRN_FEW_TASKS=500" "GPRN_FEW_TASKS=1000 GPRN_BULK_SHAPE_BIG=0"; do
env $v python bench.py --no-cpu --no-calc --blocks 4 > gpurun_out/r3_b23.json 2>gpurun_out/r3_b23.err; python -c "
import json
d=json.loads(open('gpurun_out/r3_b23.json').read().strip().splitlines()[-1]); print('[$v] cfg3:', round(d['value'],2))"
done
