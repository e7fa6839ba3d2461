#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
( while true; do sleep 60; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "whole_T500" 2>&1 | tee $OUT/e12_log.txt | tail -15
kill $HB
