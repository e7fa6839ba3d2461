# diagnostic: A/B of kernel build variants (csrc/build_variant.sh) on the default bench
run() { echo "== $1 | $2"; env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ', round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2), 'frac', round(r['frac'],3), r['kernel'])"; }
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
for A in "" "--batch 64"; do
run "PFDYN_X=0" "$A"
run "PFDYN_LIB=$V/libpfdyn_q8k.so" "$A"
run "PFDYN_X=0" "$A"
run "PFDYN_LIB=$V/libpfdyn_q8k.so" "$A"
done
