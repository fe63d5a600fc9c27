#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: top-k QPS over 10M x 768 (and chunks/s embed, when built).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: one top-k search of a query batch over this rank's corpus shard
(row-range sharding, weak scaling: `--rows` rows PER GPU), followed for N > 1 by the RCCL all-gather of
the per-shard [Q, k] results and the host-side final merge on rank 0.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).  The CPU oracle is used here only
for the `cpu_baseline` leg (timed on a bounded sample, rank 0, N = 1) -- never on the measured path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric-type", default="L2", choices=["L2", "IP", "COSINE"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also time Q in {1,16,256} (extra keys, same JSON line)")
    return ap.parse_args()


def cpu_baseline(args) -> dict:
    """The oracle (a scalar-chain C port, OpenMP over queries/rows) on a bounded sample of the workload."""
    from oracle import sc_oracle as orc

    rows, nq = 200_000, args.queries
    X = orc.synth(rows, args.dim, seed=0)
    Q = orc.synth(nq, args.dim, seed=1)
    orc.search(X[:2000], Q[:4], args.k, args.metric_type)  # warm the thread pool
    t0 = time.perf_counter()
    orc.search(X, Q, args.k, args.metric_type)
    dt = time.perf_counter() - t0
    scaled = dt * (args.rows / rows)  # exhaustive scan: linear in rows
    return {"value": nq / scaled, "unit": "queries/s", "cores": orc.threads(), "kind": "port",
            "sample": f"{rows} of {args.rows} rows x all {nq} queries in {dt:.2f}s, scaled linearly in rows"}


def main() -> None:
    args = parse()
    import numpy as np
    import torch

    from semcode_amd import _native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    stream = torch.cuda.Stream(device=dev)
    rt = _native.Runtime(device=local, stream=stream.cuda_stream)
    info = rt.device_info()

    Q, k, dim, rows = args.queries, args.k, args.dim, args.rows
    with torch.cuda.stream(stream):
        ix = _native.Index(rt, dim, metric=args.metric_type, kind="FLAT", row_base=rank * rows)
        ix.fill_synthetic(rows, seed=0, first_row=rank * rows)  # shard r = rows [r*rows, (r+1)*rows) of one global corpus
        q = torch.empty((Q, dim), dtype=torch.float32, device=dev)
        rt.synth_fill_dev(q.data_ptr(), Q, dim, dim, seed=1)
        out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
        out_r = torch.empty((Q, k), dtype=torch.int64, device=dev)
        if world > 1:
            all_d = torch.empty((world, Q, k), dtype=torch.float32, device=dev)
            all_r = torch.empty((world, Q, k), dtype=torch.int64, device=dev)
        rt.synchronize()

        def step(nq=Q):
            ix.search_dev(q.data_ptr(), nq, k, out_d.data_ptr(), out_r.data_ptr())
            if world > 1:
                dist.all_gather_into_tensor(all_d, out_d)
                dist.all_gather_into_tensor(all_r, out_r)
                if rank == 0:
                    return _native.topk_merge_host(args.metric_type, all_d.cpu().numpy(), all_r.cpu().numpy())
            return None

        def timed(nsteps, nq=Q):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                step(nq)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        for _ in range(args.warmup):
            step()
        rt.set_profiling(True)
        rt.profile_reset()
        dt = timed(args.steps)
        scan_ms, scan_n = rt.profile_read(0)
        merge_ms, merge_n = rt.profile_read(1)
        rt.set_profiling(False)

        sweep = None
        if args.sweep:
            sweep = []
            for nq in (1, 16, 256):
                for _ in range(2):
                    step(nq)
                t = timed(args.steps, nq)
                sweep.append({"queries": nq, "ms_per_batch": 1e3 * t / args.steps, "qps": nq * args.steps / t,
                              "corpus_gbs": rows * dim * 4 / (t / args.steps) / 1e9})

    ms_per_step = 1e3 * dt / args.steps
    qps = Q * args.steps / dt
    alg_bytes = rows * dim * 4  # the corpus is read once per query batch (+ norms, negligible)
    kern_ms = scan_ms / max(1, scan_n)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if scan_n else None
    line = {
        "metric": "top-k QPS over 10M x 768 (chunks/sec embed: encoder not built yet this round)",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"brute-force {args.metric_type} top-{k}, {rows} x {dim} f32 rows per GPU, batch-{Q} queries",
                   "rows_per_gpu": rows, "rows_total": rows * world, "dim": dim, "queries": Q, "k": k,
                   "metric_type": args.metric_type, "sharding": f"row-range x{world}", "device": info["name"]},
        "roofline": {"bound": "hbm", "kernel": "scan_exact_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms, "launches": scan_n,
                     "merge_avg_ms": merge_ms / max(1, merge_n)},
    }
    if sweep:
        line["sweep"] = sweep
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    ix.close()
    rt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
