#!/bin/bash
# Collect the rocprofv3 summaries of profiles/README.md for the current build (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh <tag> [bench args]   -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_hbm.csv,event_breakdown.txt}
# Counters are collected in their own passes (--pmc never together with --stats); the profiled program is python3 itself.
set -e
TAG=${1:-prof}
shift || true
ARGS="$@"
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
LIGHT="--no-cpu-baseline --no-dense-leg --no-full-trajectory --no-secondary --no-traffic"     # the profiled runs start no children of their own
python3 bench.py $ARGS $LIGHT --breakdown --steps 20 --warmup 2 > /dev/null 2> $OUT/${TAG}_event_breakdown.txt || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py $ARGS $LIGHT > $OUT/${TAG}_stats.log 2>&1
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 bench.py $ARGS $LIGHT --steps 20 --warmup 2 > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 bench.py $ARGS $LIGHT --steps 20 --warmup 2 > $OUT/${TAG}_write.log 2>&1
python3 tools/summarise_pmc.py $OUT/${TAG}_fetch $OUT/${TAG}_write > $OUT/${TAG}_pmc_hbm.csv
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write
head -8 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
cat $OUT/${TAG}_pmc_hbm.csv
cat $OUT/${TAG}_bench.json
