#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-e5}
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
timeout -k 10 900 python -m pytest tests/test_gpu_n16.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 | tee $OUT/${TAG}_log.txt || exit 1
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
bash tools/r4_cuts.sh default | tee -a $OUT/${TAG}_log.txt
PFDYN_N16=7 bash tools/r4_cuts.sh default | tee -a $OUT/${TAG}_log.txt
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
if [ -f $V/libpfdyn_sparse.so ]; then
for kid in 3 2; do
  echo "== sparse stamps KID=$kid" | tee -a $OUT/${TAG}_log.txt
  KID=$kid SHOW=3 PFDYN_LIB=$V/libpfdyn_sparse.so OFFS=0 timeout -k 10 120 python tools/n16_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/${TAG}_log.txt
done
fi
