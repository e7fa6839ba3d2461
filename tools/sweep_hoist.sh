# diagnostic: static hoist of conv layer 0 on / off and its launch policies, at several batch sizes and architectures
run() { echo "== $1 | $2"; env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ', round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2), 'frac', round(r['frac'],3), r['kernel'])"; }
for A in "--batch 128 --pharm-sizes 3-8" "--batch 512 --pharm-sizes 3-8" "--arch class-default"; do
run "PFDYN_NO_L0_HOIST=1" "$A"
run "PFDYN_X=0" "$A"
run "PFDYN_L0_RGA=1 PFDYN_L0_RGP=1" "$A"
run "PFDYN_L0_RGA=1 PFDYN_L0_RGP=2" "$A"
done
