#!/bin/bash
# experiment 2: half-wave kNN + prefetch in the tail, batched gathers in the fused launch; TAIL_CUT variant = tail without update + build
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
timeout -k 10 900 python -m pytest tests/test_gpu_n16.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 | tee $OUT/e2_log.txt || exit 1
for rep in 1 2; do
  timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e2_bench_default_$rep.json 2>> $OUT/e2_log.txt || exit 1
  PFDYN_N16=7 timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e2_bench_notail_$rep.json 2>> $OUT/e2_log.txt || exit 1
done
python - <<PY | tee -a $OUT/e2_log.txt
import json,glob
for f in sorted(glob.glob("$OUT/e2_bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j["value"]), "sample-steps/s", round(j["ms_per_step"]*1e3,2), "us/step")
    except Exception as e: print(f, "unreadable", e)
PY
for v in default cut notail; do
  unset PFDYN_LIB PFDYN_N16
  [ $v = cut ] && export PFDYN_LIB=$V/libpfdyn_cut.so
  [ $v = notail ] && export PFDYN_N16=7
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2_stats_$v -- python3 bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e2_stats_$v.log 2>&1 || exit 1
  cp $(ls $OUT/e2_stats_$v/*/*kernel_stats.csv | head -1) $OUT/e2_kernel_stats_$v.csv
  rm -rf $OUT/e2_stats_$v
  echo "== $v" | tee -a $OUT/e2_log.txt
  head -6 $OUT/e2_kernel_stats_$v.csv | cut -c1-150 | tee -a $OUT/e2_log.txt
done
