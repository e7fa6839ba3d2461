#!/bin/bash
# generic A/B: bash tools/r4_ab.sh TAG "NAME1:ENV1=V1 ENV2=V2" "NAME2:..." ...   (bench light x2 each, then rocprofv3 kernel stats each)
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=$1; shift
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
: > $OUT/${TAG}_ab.txt
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    v=$(env $envs timeout -k 10 300 python bench.py $LIGHT --steps 100 --warmup 10 2>> $OUT/${TAG}_err.txt | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step']*1e3,2))") || exit 1
    echo "$name rep$rep: $v" | tee -a $OUT/${TAG}_ab.txt
  done
done
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_$name -- python3 bench.py $LIGHT --steps 100 --warmup 10 > $OUT/ab_$name.log 2>&1 || { echo "$name rocprof failed" | tee -a $OUT/${TAG}_ab.txt; continue; }
  f=$(ls $OUT/ab_$name/*/*kernel_stats.csv | head -1)
  echo "== $name" | tee -a $OUT/${TAG}_ab.txt
  python3 - "$f" <<'PY' | tee -a $OUT/${TAG}_ab.txt
import csv,sys
tot=0
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    n=r["Name"]; n=n[n.find("k_"):][:28]
    print(f"   {n:30s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us"); tot+=float(r['AverageNs'])/1e3
print(f"   sum of the four: {tot:.2f} us")
PY
  rm -rf $OUT/ab_$name $OUT/ab_$name.log
done
