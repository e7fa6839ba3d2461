# diagnostic: hoisted layer-0 pp items on 16-row groups (PFDYN_R16_ROWS_MIN=0) against the row-group form, several batch sizes
run() { echo "== $1 | $2"; env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ', round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2), 'frac', round(r['frac'],3), r['kernel'])"; }
for A in "--batch 128 --pharm-sizes 3-8" "--batch 1024 --pharm-sizes 3-8" "--arch class-default"; do
run "PFDYN_X=0" "$A"
run "PFDYN_R16_ROWS_MIN=0" "$A"
done
