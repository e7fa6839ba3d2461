# A/B of launch-policy switches on the config-2 bench: each entry of KS_LIST is a quoted env assignment list
for cfg in "${KS_LIST[@]}"; do
  env $cfg python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-dense-leg $BENCH_ARGS 2> gpurun_out/ks_tmp.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'full_traj', round(d['full_trajectory']['value']), d['full_trajectory']['repetitions_ms'], 'edge0_us', round(d['roofline']['kernel_avg_us'],2))"
done
