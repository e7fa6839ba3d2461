#!/bin/bash
# experiment 1: stamps of the tail and fused launches; ring depth 24 A/B
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
for kid in 3 2 0; do
  echo "== stamps KID=$kid" >> $OUT/e1_stamps.txt
  KID=$kid SHOW=3 PFDYN_LIB=$V/libpfdyn_stamps.so OFFS=0 timeout -k 10 120 python tools/n16_stamps.py >> $OUT/e1_stamps.txt 2>&1 || exit 1
done
for rep in 1 2; do
  for v in default d24; do
    if [ $v = default ]; then unset PFDYN_LIB; else export PFDYN_LIB=$V/libpfdyn_$v.so; fi
    timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e1_bench_${v}_$rep.json 2>> $OUT/e1_log.txt || exit 1
    PFDYN_N16=7 timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e1_bench_${v}_notail_$rep.json 2>> $OUT/e1_log.txt || exit 1
  done
done
unset PFDYN_LIB
python - <<PY | tee -a $OUT/e1_log.txt
import json,glob
for f in sorted(glob.glob("$OUT/e1_bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j["value"]), "sample-steps/s", round(j["ms_per_step"]*1e3,2), "us/step")
    except Exception as e: print(f, "unreadable", e)
PY
cat $OUT/e1_stamps.txt
