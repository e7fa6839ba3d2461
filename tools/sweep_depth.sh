# diagnostic: A/B of kernel build variants (csrc/build_variant.sh) of the 16-row kernel
run() { echo "== $1 | $2"; env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ', round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2))"; }
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
for A in "--batch 128 --pharm-sizes 3-8" "--arch class-default"; do
run "PFDYN_X=0" "$A"
run "PFDYN_R16_ROWS_MIN=0" "$A"
run "PFDYN_R16_ROWS_MIN=0 PFDYN_LIB=$V/libpfdyn_g8.so" "$A"
run "PFDYN_R16_ROWS_MIN=0 PFDYN_LIB=$V/libpfdyn_g12.so" "$A"
done
