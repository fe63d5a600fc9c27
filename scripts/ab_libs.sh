#!/bin/bash
# Same-box A/B of two builds of the library (tuning aid): semcode_amd/_lib/libsemcode_hip.so (B) against
# semcode_amd/_lib/libsemcode_hip_base.so (A), alternating, encoder workload of bench.py.
#   gpurun -- bash scripts/ab_libs.sh
L=semcode_amd/_lib
cp $L/libsemcode_hip.so $L/new.so
for v in new base new base; do
    if [ $v = new ]; then cp $L/new.so $L/libsemcode_hip.so; else cp $L/libsemcode_hip_base.so $L/libsemcode_hip.so; fi
    timeout -k 10 200 python3 bench.py --workload embed --no-cpu-baseline --steps 20 > gpurun_out/ab_$v.log 2>&1
    python3 - "$v" <<'PY'
import json, sys
for l in open(f"gpurun_out/ab_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], round(d["value"], 1), "chunks/s  gemm frac", round(d["roofline"]["frac"], 4), " avg launch ms", round(d["roofline"]["avg_launch_ms"], 4), flush=True)
PY
done
cp $L/new.so $L/libsemcode_hip.so
