"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES CSV -> profiles/<round>_pmc_mfma.json.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d DIR -o m -- \\
        python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ivf --no-sweep
    python scripts/pmc_mfma_summary.py --csv DIR/.../m_counter_collection.csv --out profiles/r2a_pmc_mfma.json

What the counters are (MI355X_MICROARCH.md, cycle-constants table and DVFS section): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe
busy cycles summed over the SIMDs (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16); GRBM_GUI_ACTIVE is summed over the 8 XCDs,
so the kernel's elapsed shader cycles are GRBM_GUI_ACTIVE / 8 (reads high on dispatches under ~0.3 ms).  MFMA utilisation of a
launch = MFMA_BUSY / (1024 SIMDs x elapsed cycles).  `expected_busy_per_launch` (where the MFMA count of a launch is known from
its shape) cross-checks the counter's unit: the two must agree for the fraction to mean what it says."""
import argparse
import collections
import csv
import json

SIMDS = 256 * 4

ap = argparse.ArgumentParser()
ap.add_argument("--csv", required=True)
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
a = ap.parse_args()

rows = collections.defaultdict(dict)  # dispatch id -> {counter: value, name, t0, t1}
with open(a.csv, newline="") as f:
    for r in csv.DictReader(f):
        d = rows[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def short(name):
    n = name[5:] if name.startswith("void ") else name
    base = n.split("(")[0]
    return base


groups = collections.defaultdict(list)
for d in rows.values():
    groups[short(d["name"])].append(d)

# MFMAs per launch of the kernels whose shape is fixed by the bench configuration (256 chunks x 256 tokens, 10M x 768 x 1024)
M = 65536
expected = {
    "gemm256_bf16_kernel<0, 0>": 2.0 * M * 2304 * 768 / 16384 * 16,    # QKV, bias
    "gemm256_bf16_kernel<1, 0>": 2.0 * M * 3072 * 768 / 16384 * 16,    # FFN1, bias + GELU
}
res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES on `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ivf --no-sweep`, MI355X; "
                 "profiled passes clock ~3-5 % lower than unprofiled ones (MI355X_MICROARCH.md DVFS item 2); built by scripts/pmc_mfma_summary.py. " + a.note,
       "kernels": {}}
for name, ds in sorted(groups.items(), key=lambda kv: -sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in kv[1])):
    busy = sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in ds)
    if busy <= 0:
        continue
    gui = sum(x.get("GRBM_GUI_ACTIVE", 0.0) for x in ds)
    ns = sum(x["ns"] for x in ds)
    e = {"launches": len(ds), "mfma_busy_cycles_per_launch": busy / len(ds), "grbm_gui_active_per_launch": gui / len(ds),
         "elapsed_cycles_per_launch": gui / 8.0 / len(ds), "avg_ns_profiled": ns / len(ds),
         "effective_clock_ghz": (gui / 8.0) / ns if ns else None,
         "mfma_busy_frac": busy / (SIMDS * gui / 8.0) if gui else None}
    if "SQ_WAVE_CYCLES" in ds[0]:
        e["sq_wave_cycles_per_launch_x4"] = 4.0 * sum(x.get("SQ_WAVE_CYCLES", 0.0) for x in ds) / len(ds)  # counted in quad-cycles
    for k, v in expected.items():
        if name.startswith(k.split("<")[0]) and k.split("<")[1].split(",")[0] == (name.split("<")[1].split(",")[0] if "<" in name else ""):
            e["expected_busy_per_launch"] = v
    res["kernels"][name] = e
with open(a.out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
