#!/bin/bash
# Round-4 GPU check (run through gpurun): targeted tests first, then the whole -m gpu suite, then A/B bench + rocprofv3 kernel stats.
#   bash tools/r4_check.sh <tag> [quick]
set -o pipefail
TAG=${1:-r4}
MODE=${2:-full}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
echo "== targeted tests" | tee $OUT/${TAG}_log.txt
timeout -k 10 600 python -m pytest tests/test_gpu_n16.py -x -q -k "tail" 2>&1 | tail -15 | tee -a $OUT/${TAG}_log.txt || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "trajectory" 2>&1 | tail -15 | tee -a $OUT/${TAG}_log.txt || exit 1
if [ "$MODE" = "full" ]; then
  echo "== full gpu suite" | tee -a $OUT/${TAG}_log.txt
  timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/${TAG}_log.txt || exit 1
fi
echo "== bench A/B" | tee -a $OUT/${TAG}_log.txt
for rep in 1 2; do
  timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/${TAG}_bench_default_$rep.json 2>> $OUT/${TAG}_log.txt || exit 1
  PFDYN_N16=7 timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 > $OUT/${TAG}_bench_notail_$rep.json 2>> $OUT/${TAG}_log.txt || exit 1
done
python - <<PY | tee -a $OUT/${TAG}_log.txt
import json,glob
for f in sorted(glob.glob("$OUT/${TAG}_bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j["value"]), "sample-steps/s", round(j["ms_per_step"]*1e3,2), "us/step")
    except Exception as e: print(f, "unreadable", e)
PY
echo "== rocprofv3 kernel stats (default policy)" | tee -a $OUT/${TAG}_log.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py $LIGHT --steps 100 --warmup 10 > $OUT/${TAG}_stats.log 2>&1 || exit 1
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_stats
head -8 $OUT/${TAG}_kernel_stats.csv | cut -c1-200 | tee -a $OUT/${TAG}_log.txt
