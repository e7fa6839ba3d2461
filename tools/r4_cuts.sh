#!/bin/bash
# per-phase kernel durations from cut variants (diagnostic builds that end a workgroup at a phase boundary): rocprofv3 averages
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
export TMPDIR=/tmp
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
: > $OUT/cuts.txt
for v in "$@"; do
  if [ $v = default ]; then unset PFDYN_LIB; else export PFDYN_LIB=$V/libpfdyn_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cut_$v -- python3 bench.py $LIGHT --steps 100 --warmup 10 > $OUT/cut_$v.log 2>&1 || { echo "$v failed" >> $OUT/cuts.txt; tail -3 $OUT/cut_$v.log >> $OUT/cuts.txt; continue; }
  f=$(ls $OUT/cut_$v/*/*kernel_stats.csv | head -1)
  echo "== $v" >> $OUT/cuts.txt
  python3 - "$f" >> $OUT/cuts.txt <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    n=r["Name"]; n=n[n.find("k_"):][:28]
    print(f"   {n:30s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
  rm -rf $OUT/cut_$v $OUT/cut_$v.log
done
cat $OUT/cuts.txt
