#!/bin/bash
# A/B line for the throughput regime of the row-group kernels: config 3, class-default depth, batch 1024, plus config 2
C="--no-cpu-baseline --no-dense-leg --no-full-trajectory"
for cfg in "--batch 32" "--batch 128 --pharm-sizes 3-8" "--batch 128 --arch class-default" "--batch 1024 --pharm-sizes 3-8 --steps 40" "--batch 32 --arch class-default"; do
  python3 bench.py $cfg $C 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg:', round(d['value']), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_avg_us'],1), round(r['frac'],3), round(r.get('frac_executed',0),3))"
done
