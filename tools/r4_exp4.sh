#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p $OUT
V=$PWD/pharmacophore-diffusion_amd/csrc/variants
: > $OUT/e4_stamps.txt
for kid in 3 2 0; do
  echo "== sparse stamps KID=$kid" >> $OUT/e4_stamps.txt
  KID=$kid SHOW=4 PFDYN_LIB=$V/libpfdyn_sparse.so OFFS=0 timeout -k 10 120 python tools/n16_stamps.py >> $OUT/e4_stamps.txt 2>&1 || exit 1
done
cat $OUT/e4_stamps.txt
