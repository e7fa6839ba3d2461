# diagnostic: A/B of kernel build variants (csrc/build_variant.sh NAME "-D...") on bench.py
#   VARIANTS="name1 name2" ARGS="--batch 128 --pharm-sizes 3-8" ENVS="PFDYN_R16_ROWS_MIN=0" bash tools/sweep_depth.sh
run() { echo "== $1 | $2"; env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   ', round(d['value']), 'ms/step', round(d['ms_per_step'],5), 'edge_us', round(r['kernel_avg_us'],2), 'frac', round(r['frac'],3), r['kernel'])"; }
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
E=${ENVS:-PFDYN_X=0}
run "$E" "$ARGS"
for v in $VARIANTS; do run "$E PFDYN_LIB=$V/libpfdyn_$v.so" "$ARGS"; done
