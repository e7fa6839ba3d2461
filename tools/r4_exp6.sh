#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-e8}
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
timeout -k 10 900 python -m pytest tests/test_gpu_n16.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 | tee $OUT/${TAG}_log.txt || exit 1
for rep in 1 2; do
  timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/${TAG}_bench_default_$rep.json 2>> $OUT/${TAG}_log.txt || exit 1
  PFDYN_XCD_SPLIT=1 timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/${TAG}_bench_nosplit_$rep.json 2>> $OUT/${TAG}_log.txt || exit 1
done
python - <<PY | tee -a $OUT/${TAG}_log.txt
import json,glob
for f in sorted(glob.glob("$OUT/${TAG}_bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j["value"]), "sample-steps/s", round(j["ms_per_step"]*1e3,2), "us/step")
    except Exception as e: print(f, "unreadable", e)
PY
bash tools/r4_cuts.sh default | tee -a $OUT/${TAG}_log.txt
PFDYN_XCD_SPLIT=1 bash tools/r4_cuts.sh default | tee -a $OUT/${TAG}_log.txt
