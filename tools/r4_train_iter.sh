#!/bin/bash
# training-step iteration: the training tests, then the step with and without a kernel trace (timeline of one step)
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
( while true; do sleep 60; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
timeout -k 10 900 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_api.py -m gpu -x -q -k "not two_rank and not bench" > $OUT/t_train.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $OUT/t_train.log
if [ $rc -eq 0 ]; then
python3 bench.py --train --steps 100 --warmup 10 > $OUT/train_bench.json 2>/dev/null; echo "train rc=$?"
python3 bench.py --train --steps 100 --warmup 10 --train-batches 1 > $OUT/train_fixed_batch_bench.json 2>/dev/null
python3 bench.py --train --train-dtype bf16 --steps 100 --warmup 10 > $OUT/train_bf16_bench.json 2>/dev/null; echo "train bf16 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 bench.py --train --steps 20 --warmup 4 > $OUT/train_stats.log 2>&1
cp $(ls $OUT/train_stats/*/*kernel_stats.csv | head -1) $OUT/train_kernel_stats.csv
python3 tools/train_step_timeline.py $OUT/train_stats > $OUT/train_step_timeline.txt 2>&1; rm -rf $OUT/train_stats
for f in train_bench train_fixed_batch_bench train_bf16_bench; do python3 -c "import json,sys; j=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(j['value']), round(j['ms_per_step'],4))"; done
fi
kill $HB
