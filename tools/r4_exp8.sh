#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
( while true; do sleep 60; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_r4a.json 2> $OUT/bench_r4a.err; echo "bench rc=$?"
python3 - <<PY
import json
j=json.loads(open("$OUT/bench_r4a.json").read().strip().splitlines()[-1])
print(round(j["value"]), j["ms_per_step"]*1e3, "us/step")
r=j["roofline"]; print(r["kernel"][:60], r["kernel_avg_us"], r["frac"], r["frac_executed"], r.get("launches_sum_us"))
for e in r.get("launches", []): print("   ", e["kernel"][:40], e["launches_per_step"], round(e["avg_us"],2), e["frac"], e["frac_executed"], round(e["fetch_bytes"]), round(e["write_bytes"]))
print(r["conv_layer0_launch"])
print(j.get("secondary",{}).keys())
PY
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/e13_tests.txt | tail -12
kill $HB
