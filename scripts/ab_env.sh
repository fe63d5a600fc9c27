#!/bin/bash
# Same-box A/B of runtime-selectable variants (environment switches read by the library), encoder workload of bench.py,
# alternating so that clock / thermal drift hits every arm:   gpurun -- bash scripts/ab_env.sh "SC_GEMM_VAR=0" "SC_GEMM_VAR=32" ...
for round in 1 2; do
  for arm in "$@"; do
    env $arm timeout -k 10 200 python3 bench.py --workload embed --no-cpu-baseline --steps 20 > gpurun_out/ab_env.log 2>&1
    python3 - "$arm" <<'PY'
import json, sys
for l in open("gpurun_out/ab_env.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], round(d["value"], 1), "chunks/s  gemm frac", round(d["roofline"]["frac"], 4), " avg launch ms", round(d["roofline"]["avg_launch_ms"], 4), flush=True)
PY
  done
done
