#!/bin/bash
export TMPDIR=/tmp
for f in 8 16 32 4; do
  export PFDYN_REDUCE_FORM=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trs_$f -- python3 bench.py --train --steps 20 --warmup 4 > gpurun_out/trs.log 2>&1
  echo "form $f: $(grep -E 'k_train_reduce' $(ls gpurun_out/trs_$f/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | tr '\n' ' ')"
  rm -rf gpurun_out/trs_$f
  python3 bench.py --train --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('   step', round(j['value']), round(j['ms_per_step'],4))"
done
