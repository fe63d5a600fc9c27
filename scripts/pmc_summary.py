"""rocprofv3 --pmc CSVs -> profiles/<round>_pmc_traffic.json (the `traffic` figures bench.py reports).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR_F -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR_W -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR_E -o e -- python3 bench.py --workload scan --queries 16 --steps 3 --warmup 1 --no-cpu-baseline
    python scripts/pmc_summary.py --fetch DIR_F/.../f_counter_collection.csv --write DIR_W/.../w_counter_collection.csv \\
        --fetch-exact DIR_E/.../e_counter_collection.csv --out profiles/r1r_pmc_traffic.json

Counter unit: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B for each 128-B request of a wide
coalesced streaming read (16 B/lane, global_load and LDS-DMA alike), so it is doubled (MI355X_MICROARCH.md, HBM section).
Separate passes per counter, kernel-trace/stats in yet another run, as the same guide prescribes."""
import argparse
import collections
import csv
import hashlib
import json
import os
from pathlib import Path


def csrc_sha() -> str:
    """the same hash bench.py computes (bench.csrc_sha): ties these figures to the kernel sources they were measured on"""
    h = hashlib.sha256()
    for f in sorted((Path(__file__).resolve().parent.parent / "semcode_amd" / "csrc").iterdir()):
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def per_kernel(path):
    d = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return d


def pick(d, prefix):
    out = []
    for k, v in d.items():
        if k.startswith(prefix) or k.startswith("void " + prefix):
            out += v
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--fetch-exact")
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
a = ap.parse_args()
F, W = per_kernel(a.fetch), per_kernel(a.write)
res = {"csrc_sha": csrc_sha(), "commit": os.environ.get("SC_COMMIT", "?"),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 1 --no-cpu-baseline`, "
                 "MI355X; FETCH_SIZE doubled for 16 B/lane streaming reads as MI355X_MICROARCH.md prescribes; built by scripts/pmc_summary.py. " + a.note}
g_f, g_w = pick(F, "gemm256_bf16_kernel"), pick(W, "gemm256_bf16_kernel")
if g_f and g_w:
    fx2, wr = 2.0 * sum(g_f) / len(g_f), sum(g_w) / len(g_w)
    res["gemm256_bf16_kernel"] = {"fetch_bytes_x2_per_launch": fx2, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": fx2 + wr,
                                  "launches": len(g_f), "algorithmic_bytes_per_launch_avg": 455500000.0,
                                  "per_epilogue": {k.split("<")[1].split(",")[0]: {"fetch_x2": 2.0 * sum(v) / len(v), "launches": len(v)}
                                                   for k, v in F.items() if "gemm256_bf16_kernel<" in k}}
steps = len(pick(F, "scan_rerank_kernel"))
c_f, c_w = pick(F, "scan_coarse256"), pick(W, "scan_coarse256")  # scan_coarse256_kernel (dense phases) + scan_coarse256p_kernel (persistent)
if steps and c_f:
    res["scan_coarse256_kernel"] = {"steps": steps, "launches_per_step": len(c_f) / steps, "fetch_bytes_x2_per_step": 2.0 * sum(c_f) / steps,
                                    "write_bytes_per_step": (sum(c_w) / steps) if c_w else None}
for name in ("layernorm_kernel", "attention_kernel", "scan_rerank_kernel", "scan_select_kernel", "mean_pool_kernel", "embed_ln_kernel"):
    f, w = pick(F, name), pick(W, name)
    if f:
        res[name] = {"fetch_bytes_x2_per_launch": 2.0 * sum(f) / len(f), "write_bytes_per_launch": (sum(w) / len(w)) if w else None, "launches": len(f)}
if a.fetch_exact:
    e = pick(per_kernel(a.fetch_exact), "scan_exact_kernel")
    if e:
        res["scan_exact_kernel"] = {"fetch_bytes_x2_per_launch": 2.0 * sum(e) / len(e), "launches": len(e)}
with open(a.out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
