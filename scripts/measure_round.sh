#!/bin/bash
# One measurement pass for profiles/: bench line, rocprofv3 kernel stats, PMC traffic (separate passes).  Run on the GPU box:
#   gpurun -- bash scripts/measure_round.sh r1r
set -e -o pipefail
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/measure_$tag
mkdir -p "$out"
timeout -k 10 500 python3 bench.py > "$out/bench.log" 2>&1
echo "bench done" && tail -c 300 "$out/bench.log"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o "$tag" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/stats.log" 2>&1
echo "stats done"
# counter passes: one counter family per run, no ivf / sweep (thousands of serialised launches under PMC)
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ivf --no-sweep"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f" -o f -- python3 bench.py $PMC_ARGS > "$out/pmc_f.log" 2>&1
echo "pmc fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w" -o w -- python3 bench.py $PMC_ARGS > "$out/pmc_w.log" 2>&1
echo "pmc write done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_e" -o e -- python3 bench.py --workload scan --queries 16 --search-mode exact $PMC_ARGS > "$out/pmc_e.log" 2>&1
echo "pmc exact done"
if timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$out/pmc_m" -o m -- python3 bench.py $PMC_ARGS > "$out/pmc_m.log" 2>&1; then
    python3 scripts/pmc_mfma_summary.py --csv "$(find "$out/pmc_m" -name '*counter_collection.csv' | head -1)" --out "$out/${tag}_pmc_mfma.json" > /dev/null && echo "pmc mfma done"
else
    echo "pmc mfma pass FAILED (see pmc_m.log)"; tail -5 "$out/pmc_m.log"
fi
python3 scripts/pmc_summary.py --fetch "$(find "$out/pmc_f" -name '*counter_collection.csv' | head -1)" --write "$(find "$out/pmc_w" -name '*counter_collection.csv' | head -1)" \
    --fetch-exact "$(find "$out/pmc_e" -name '*counter_collection.csv' | head -1)" --out "$out/${tag}_pmc_traffic.json" > /dev/null
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
# keep the merged-back directory small: the raw traces are not needed
find "$out" -name '*.db' -delete; find "$out" -name '*kernel_trace.csv' -delete; find "$out" -name '*agent_info.csv' -delete
ls -la "$out"
