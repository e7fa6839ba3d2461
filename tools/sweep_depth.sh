# diagnostic: sensitivity of the step to the depth of the quad prefetch ring (variants built by csrc/build_variant.sh)
run() { echo "== $1"; env $1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2), 'frac', round(r['frac'],3), r['kernel'])"; }
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
run "PFDYN_X=0"
run "PFDYN_LIB=$V/libpfdyn_d2_24.so"
run "PFDYN_LIB=$V/libpfdyn_d1_12.so PFDYN_RG2_ROWS_MIN=1000000000"
run "PFDYN_RG2_ROWS_MIN=1000000000"
