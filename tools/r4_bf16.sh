#!/bin/bash
# bf16 training leg: gradient contract at batch 256, step time and per-kernel durations of both precisions
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/bf16_grad_check.py 256 > $OUT/bf16_check256.txt 2>&1; tail -8 $OUT/bf16_check256.txt
for d in f32 bf16; do
  python3 bench.py --train --train-dtype $d --steps 100 --warmup 10 2>/dev/null | tail -1 > $OUT/train_${d}_bench.json
  python3 -c "import json; j=json.load(open('$OUT/train_${d}_bench.json')); print('$d', round(j['value']), round(j['ms_per_step'],4), j['final_loss'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$d -- python3 bench.py --train --train-dtype $d --steps 20 --warmup 4 > $OUT/tr_$d.log 2>&1
  cp $(ls $OUT/tr_$d/*/*kernel_stats.csv | head -1) $OUT/train_${d}_kernel_stats.csv; rm -rf $OUT/tr_$d
  head -14 $OUT/train_${d}_kernel_stats.csv | cut -c1-150
done
