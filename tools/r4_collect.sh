#!/bin/bash
# Round-4 profile collection (run through gpurun; copies to keep go to profiles/r04/ afterwards)
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
( while true; do sleep 60; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"
bash tools/collect_profiles.sh c2 > $OUT/c2_collect.log 2>&1; echo "c2 rc=$?"
bash tools/collect_profiles.sh c3 --batch 128 --pharm-sizes 3-8 --no-secondary > $OUT/c3_collect.log 2>&1; echo "c3 rc=$?"
bash tools/collect_profiles.sh cd --arch class-default --no-secondary > $OUT/cd_collect.log 2>&1; echo "cd rc=$?"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 $LIGHT 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('driver-style bare', round(j['value']), round(j['ms_per_step']*1e3,2))"; done | tee $OUT/c2_driver_style_repeats.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c2_driver_style_bench.json 2>/dev/null; echo "driver rc=$?"
python3 bench.py --train --steps 100 --warmup 10 > $OUT/train_bench.json 2>/dev/null; echo "train rc=$?"
python3 bench.py --train --steps 100 --warmup 10 --train-batches 1 > $OUT/train_fixed_batch_bench.json 2>/dev/null
python3 bench.py --train --train-dtype bf16 --steps 100 --warmup 10 > $OUT/train_bf16_bench.json 2>/dev/null; echo "train bf16 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bf16_stats -- python3 bench.py --train --train-dtype bf16 --steps 20 --warmup 4 > $OUT/train_bf16_stats.log 2>&1
cp $(ls $OUT/train_bf16_stats/*/*kernel_stats.csv | head -1) $OUT/train_bf16_kernel_stats.csv; rm -rf $OUT/train_bf16_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -- python3 bench.py --train --steps 20 --warmup 4 > $OUT/train_stats.log 2>&1
cp $(ls $OUT/train_stats/*/*kernel_stats.csv | head -1) $OUT/train_kernel_stats.csv; python3 tools/train_step_timeline.py $OUT/train_stats > $OUT/train_step_timeline.txt 2>&1; rm -rf $OUT/train_stats
python3 bench.py --sample-slice 64 > $OUT/config4_slice_bench.json 2>/dev/null; echo "slice rc=$?"
python3 bench.py --sample-slice 1000 > $OUT/config4_full_bench.json 2>/dev/null; echo "full rc=$?"
# the optional tail launches, both forms, next to the default
bash tools/r4_ab.sh tailforms "merged:PFDYN_X=1" "separate:PFDYN_HS_BUILD=0" "tail_rg:PFDYN_N16=15 PFDYN_TAIL_FORM=rg" "tail_n16:PFDYN_N16=15 PFDYN_TAIL_FORM=n16" > $OUT/tail_forms.txt 2>&1
kill $HB
tail -3 $OUT/c2_collect.log
