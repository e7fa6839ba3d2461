#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
timeout -k 10 900 python -m pytest tests/test_gpu_n16.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 | tee $OUT/e3_log.txt || exit 1
for rep in 1 2; do
  timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e3_bench_default_$rep.json 2>> $OUT/e3_log.txt || exit 1
  PFDYN_N16=7 timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/e3_bench_notail_$rep.json 2>> $OUT/e3_log.txt || exit 1
done
python - <<PY | tee -a $OUT/e3_log.txt
import json,glob
for f in sorted(glob.glob("$OUT/e3_bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j["value"]), "sample-steps/s", round(j["ms_per_step"]*1e3,2), "us/step")
    except Exception as e: print(f, "unreadable", e)
PY
bash tools/r4_cuts.sh default t3 b8 b10 | tee -a $OUT/e3_log.txt
export PFDYN_N16=7
bash tools/r4_cuts.sh default | tee -a $OUT/e3_log.txt
