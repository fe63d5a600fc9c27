#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: chunks/s embed (256-token chunks) + top-k QPS over 10M x 768.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Two workloads, both with
inputs already resident in HBM when the timed region starts:

  embed (BASELINE configs[1], the primary `value`): one step = the transformer-encoder forward over a
      batch of 256 chunks x 256 token ids (BERT-base shape, bf16 MFMA, random-init weights), pooled
      f32 vectors out.  Pure data parallel: every rank embeds its own batch, no collective;
      value = chunks of all ranks / max-over-ranks time.
  scan (BASELINE configs[2]/[3], reported under "topk"): one step = brute-force L2 top-10 of a
      batch of 1024 queries over this rank's 10M x 768 f32 shard (row-range sharding, weak scaling),
      then for N > 1 the RCCL all-gather of the per-shard [Q, k] results and the host-side merge on rank 0.

Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` legs (bounded samples,
rank 0, N = 1) -- never on the measured path.
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

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
MFMA_I8_PEAK_TOPS = 5000.0      # MI355X_MICROARCH.md, matrix-core table: i8 16x16x64 = the cycles of bf16 16x16x32 at twice the K


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="both", choices=["both", "embed", "scan"], help="both = embed + scan (+ the IVF_FLAT block at N = 1)")
    # embed
    ap.add_argument("--batch", type=int, default=256, help="chunks per step per GPU")
    ap.add_argument("--seq", type=int, default=256, help="tokens per chunk")
    # scan
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric-type", default="L2", choices=["L2", "IP", "COSINE"])
    ap.add_argument("--search-mode", default="auto", choices=["auto", "exact", "batched"], help="scan: path of the timed step (exact: the f32 scan kernel whatever the batch size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="scan: skip the Q in {1, 16, 32, 256} sweep (SURVEY 8d: report 1, 32, 256, 1024; 16 is the exact kernel's largest single pass)")
    ap.add_argument("--sweep", action="store_true", help="(default now; kept for old command lines)")
    # ivf (BASELINE configs[4]): IVF_FLAT nlist 4096 / nprobe 64 over 10M x 3072, N = 1 only
    ap.add_argument("--no-ivf", action="store_true")
    ap.add_argument("--ivf-rows", type=int, default=10_000_000)
    ap.add_argument("--ivf-dim", type=int, default=3072)
    ap.add_argument("--ivf-nlist", type=int, default=4096)
    ap.add_argument("--ivf-nprobe", type=int, default=64)
    return ap.parse_args()


class Ctx:
    pass


def csrc_sha() -> str:
    """sha256 over the kernel sources of the library this process runs (semcode_amd/csrc/*.{hip,h,cpp}, names and contents):
    what scripts/pmc_summary.py stamps into a PMC summary, so that a measured traffic figure can be tied to the code it was
    measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / "semcode_amd" / "csrc").iterdir()):
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


_PMC = {}


def pmc_traffic():
    """HBM-side bytes from the last committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on
    gfx950 for 16 B/lane streaming reads).  PMC counters cannot be read from inside this process, so these are
    the committed figures of the same kernels on the same workload -- and only when the summary was measured on THIS
    library's kernel sources (its "csrc_sha" equals csrc_sha()): otherwise every traffic field is null and the note
    says why.  None when no summary is committed."""
    if "d" in _PMC:
        return _PMC["d"]
    d = None
    try:
        latest = sorted((ROOT / "profiles").glob("*_pmc_traffic.json"))[-1]  # named per round: the newest one
        raw = json.loads(latest.read_text())
        have, want = raw.get("csrc_sha"), csrc_sha()
        if have == want:
            d = raw
            d["_file"] = f"profiles/{latest.name} (measured on csrc {have}, commit {raw.get('commit', '?')})"
        else:
            d = {"_file": f"profiles/{latest.name} was measured on csrc {have or 'unstamped'}, this library is {want}: traffic not reported (re-run scripts/measure_round.sh)"}
    except Exception:
        d = None
    _PMC["d"] = d
    return d


def timed(ctx, fn, nsteps):
    """EXACTLY nsteps calls of fn, bracketed by barrier + synchronize on both sides; max over ranks."""
    import torch

    def barrier():
        if ctx.comm is not None:
            ctx.comm.allreduce_max(0.0)  # an RCCL all-reduce on the runtime's stream + its synchronisation
        else:
            ctx.dist.barrier()

    torch.cuda.synchronize()
    if ctx.world > 1:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        fn()
    if getattr(ctx, "drain", None):
        ctx.drain()  # rank 0: the last step's host merge
    torch.cuda.synchronize()
    if ctx.world > 1:
        barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ctx.world > 1:
        if ctx.comm is not None:
            dt = ctx.comm.allreduce_max(dt)
        else:
            t = torch.tensor([dt], dtype=torch.float64)
            ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)  # control group (gloo): a host double
            dt = float(t.item())
    return dt


# ------------------------------------------------------------------------------------------ embed

def encoder_flops_per_chunk(S: int, H: int = 768, L: int = 12, F: int = 3072):
    gemm = L * 2 * S * H * (3 * H + H + 2 * F)   # QKV + out-proj + FFN1 + FFN2
    attn = L * 4 * S * S * H                     # QK^T + PV
    return gemm, attn                            # S=256: 4.349e10 + 2.416e9 = 4.590e10


def bench_embed(ctx, args) -> dict:
    import torch

    from semcode_amd import _native

    B, S = args.batch, args.seq
    enc = _native.Encoder(ctx.rt, dict(_native.BERT_BASE), weights=None, synth_seed=0)
    g = torch.Generator(device="cpu").manual_seed(1 + ctx.rank)
    ids = torch.randint(1000, 30000, (B, S), generator=g, dtype=torch.int32).to(ctx.dev)
    lens = torch.full((B,), S, dtype=torch.int32, device=ctx.dev)
    out = torch.empty((B, 768), dtype=torch.float32, device=ctx.dev)

    def step():
        enc.embed_ids_dev(ids.data_ptr(), lens.data_ptr(), B, S, out.data_ptr())

    for _ in range(args.warmup):
        step()
    # hipEvent pairs around every 5th GEMM / attention launch: 5 is co-prime with the 4 GEMM shapes of a layer, so all shapes are
    # sampled equally; pairs around all ~60 launches of a step cost 1.6 % of it (SEMCODE_BENCH_NOPROF=1 measures without any)
    ctx.rt.set_profiling(0 if os.environ.get("SEMCODE_BENCH_NOPROF") else 5)
    ctx.rt.profile_reset()
    dt = timed(ctx, step, args.steps)
    gemm_ms, gemm_n = ctx.rt.profile_read(2)
    attn_ms, attn_n = ctx.rt.profile_read(3)
    ctx.rt.set_profiling(False)
    assert bool(torch.isfinite(out).all()), "encoder produced non-finite values"
    enc.close()

    gemm_fl, attn_fl = encoder_flops_per_chunk(S)
    chunks_s = B * ctx.world * args.steps / dt
    # sampled launches cover the 4 shapes equally, so the mean flops of a sampled launch = the mean over all 48 per step
    gemm_tflops = (gemm_fl * B / 48 * gemm_n) / (gemm_ms * 1e-3) / 1e12 if gemm_n else None
    res = {
        "chunks_per_s": chunks_s,
        "ms_per_step": 1e3 * dt / args.steps,
        "workload": f"encoder forward, {B} chunks x {S} tokens per GPU per step, BERT-base shape (12x768, 12 heads, FFN 3072), bf16 MFMA / f32 accumulate, random-init weights",
        "roofline": {"bound": "mfma", "kernel": "gemm256_bf16_kernel (LayerNorm-folded epilogues: the launches also do the two LayerNorms per layer)", "achieved": gemm_tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / MFMA_BF16_PEAK_TFLOPS if gemm_tflops else None,
                     "traffic": (pmc_traffic() or {}).get("gemm256_bf16_kernel", {}).get("traffic_bytes_per_launch") if (B, S) == (256, 256) else None,
                     "traffic_note": f"bytes per launch (avg of the 4 GEMM shapes), {(pmc_traffic() or {}).get('_file')}; algorithmic A+W+C(+R) = 4.56e8",
                     "algorithmic_flops_per_launch": gemm_fl * B / 48, "avg_launch_ms": gemm_ms / max(1, gemm_n), "launches": 48 * args.steps, "launches_timed": gemm_n,
                     "attention_avg_launch_ms": attn_ms / max(1, attn_n),
                     "end_to_end_tflops": chunks_s / ctx.world * (gemm_fl + attn_fl) / 1e12,
                     "end_to_end_frac": chunks_s / ctx.world * (gemm_fl + attn_fl) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
    }
    return res


def cpu_baseline_embed(args) -> dict:
    """torch-CPU f32 restatement (oracle/bert_torch.py: one sgemm over all token rows per layer matmul, threaded element-wise
    ops, all host cores) on a bounded sample of the same workload: one full batch when the box is large, never < 32 chunks."""
    import numpy as np

    from oracle import bert_oracle as bo
    from oracle import bert_torch as bt

    from oracle.sc_oracle import host_cores

    cores = host_cores()  # affinity mask capped by the cgroup CPU quota
    cfg = dict(bo.BERT_BASE)
    blob = bo.make_blob(cfg, 0, "bench")
    n = args.batch if cores >= 64 else 64
    rng = np.random.default_rng(1)
    ids = rng.integers(1000, 30000, size=(n, args.seq)).astype(np.int32)
    lens = np.full(n, args.seq, np.int32)
    bt.forward(cfg, blob, ids[:8], lens[:8], threads=cores)
    t0 = time.perf_counter()
    bt.forward(cfg, blob, ids, lens, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "chunks/s", "cores": cores, "kind": "port",
            "sample": f"{n} chunks x {args.seq} tokens in one call, torch-CPU f32 restatement of the same forward (BLAS + threaded ops on all cores), {dt:.2f}s"}


# ------------------------------------------------------------------------------------------- scan

def bench_scan(ctx, args) -> dict:
    import torch

    from semcode_amd import _native

    Q, k, dim, rows = args.queries, args.k, args.dim, args.rows
    rt, dev, world, rank = ctx.rt, ctx.dev, ctx.world, ctx.rank
    ix = _native.Index(rt, dim, metric=args.metric_type, kind="FLAT", row_base=rank * rows)
    ix.fill_synthetic(rows, seed=0, first_row=rank * rows)  # shard r = rows [r*rows, (r+1)*rows) of ONE global corpus
    if args.search_mode != "auto":
        ix.set_search_mode(args.search_mode)
    q = torch.empty((Q, dim), dtype=torch.float32, device=dev)
    rt.synth_fill_dev(q.data_ptr(), Q, dim, dim, seed=1)
    out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
    out_r = torch.empty((Q, k), dtype=torch.int64, device=dev)
    if world > 1:
        all_d = torch.empty((world, Q, k), dtype=torch.float32, device=dev)
        all_r = torch.empty((world, Q, k), dtype=torch.int64, device=dev)
        if ctx.comm is not None:
            from semcode_amd.storage.sharded import ShardedSearcher

            searcher = ShardedSearcher(ix, args.metric_type, comm=ctx.comm)
    rt.synchronize()
    # The host-side final merge (north_star) runs on rank 0 only.  It is taken off the step's critical path: a worker thread merges
    # step i's gathered lists (sc_topk_merge_host releases the GIL) while the device scans step i + 1; timed() drains it before
    # the clock stops (merge_drain), and its own time is reported (host_merge_ms_per_step).
    from concurrent.futures import ThreadPoolExecutor

    merger = ThreadPoolExecutor(max_workers=1) if world > 1 and rank == 0 else None
    pending = []
    merge_time = [0.0, 0]

    def host_merge(gd, gr):
        t0 = time.perf_counter()
        out = _native.topk_merge_host(args.metric_type, gd, gr)
        merge_time[0] += time.perf_counter() - t0
        merge_time[1] += 1
        return out

    def submit_merge(gd, gr):
        while pending:  # at most one merge in flight behind the step that produced it
            pending.pop().result()
        pending.append(merger.submit(host_merge, gd.cpu().numpy(), gr.cpu().numpy()))

    def merge_drain():
        while pending:
            pending.pop().result()

    ctx.drain = merge_drain if merger else None

    def step(nq=Q):
        if world > 1 and ctx.comm is not None:
            # the product path (storage.sharded.ShardedSearcher.search_dev): shard search + the ONE exchange step (RCCL all-gather
            # of every shard's [Q,k] over xGMI, issued by libsemcode_hip on its stream), then the host merge on rank 0
            searcher.search_dev(q.data_ptr(), nq, k, all_d.data_ptr(), all_r.data_ptr())
            if rank == 0:
                gd = all_d.view(-1)[: world * nq * k].view(world, nq, k)  # the gathered arrays are [world, nq, k] from the buffers' start
                gr = all_r.view(-1)[: world * nq * k].view(world, nq, k)
                submit_merge(gd, gr)
            return None
        ix.search_dev(q.data_ptr(), nq, k, out_d.data_ptr(), out_r.data_ptr())
        if world > 1:  # no native communicator: torch.distributed carries the exchange
            if ctx.nccl_group is not None:
                ctx.dist.all_gather_into_tensor(all_d, out_d, group=ctx.nccl_group)
                ctx.dist.all_gather_into_tensor(all_r, out_r, group=ctx.nccl_group)
                gd, gr = all_d, all_r
            else:  # rehearsal backend: gather on the host
                hd, hr = out_d.cpu(), out_r.cpu()
                ld, lr = [torch.empty_like(hd) for _ in range(world)], [torch.empty_like(hr) for _ in range(world)]
                ctx.dist.all_gather(ld, hd)
                ctx.dist.all_gather(lr, hr)
                gd, gr = torch.stack(ld), torch.stack(lr)
            if rank == 0:
                submit_merge(gd, gr)
        return None

    for _ in range(args.warmup):
        step()
    merge_drain()
    merge_time[0], merge_time[1] = 0.0, 0
    rt.set_profiling(True)
    rt.profile_reset()
    dt = timed(ctx, step, args.steps)
    scan_ms, scan_n = rt.profile_read(0)
    merge_ms, merge_n = rt.profile_read(1)
    rt.set_profiling(False)
    st = ix.last_search_stats()
    path, unc = st["path"], st["uncertified"]
    coarse_bits, handed = st.get("coarse_bits", 16), st.get("handed_to_bf16", 0)
    groups = (Q + 15) // 16  # exact path: one corpus pass per 16 queries

    # Batch-size sweep: the scan is HBM-bound for small batches (exact f32 kernel, one corpus pass per <= 16 queries)
    # and MFMA-bound at Q = 1024.  The exact kernel's own launches are bracketed by hipEvents like the coarse kernel's above.
    sweep, exact_roof = None, None
    if not args.no_sweep:
        sweep = []
        # (1 and 16 queries also with the exact f32 scan forced: the HBM roofline block below is that kernel's; the planner itself
        # answers such batches through the int8 narrow kernel where an int8 shadow may exist)
        for nq, mode in ((1, "auto"), (16, "auto"), (32, "auto"), (256, "auto"), (1, "exact"), (16, "exact")):
            if nq > Q:
                continue
            ix.set_search_mode(mode)
            for _ in range(2):
                step(nq)
            rt.set_profiling(True)
            rt.profile_reset()
            t = timed(ctx, lambda: step(nq), args.steps)
            k_ms, k_n = rt.profile_read(0)
            rt.set_profiling(False)
            stq = ix.last_search_stats()
            p = stq["path"]
            # bytes the scan kernels actually stream per batch: the f32 rows (exact path: one pass per 16 queries), or the coarse shadow
            # (batched path: int8 = 1 B, bf16 = 2 B per padded element)
            ldq = (dim + 63) // 64 * 64
            streamed = rows * ldq * 4 * ((nq + 15) // 16) if p == "exact" else rows * ((ldq + 127) // 128 * 128 if stq.get("coarse_bits", 16) == 8 else ldq * 2)
            sweep.append({"queries": nq, "mode": mode, "path": p, "ms_per_batch": 1e3 * t / args.steps, "qps": nq * args.steps * world / t,
                          "scan_kernel_ms_per_batch": k_ms / args.steps,
                          "algorithmic_f32_gbs": rows * dim * 4 / (t / args.steps) / 1e9,  # SURVEY 8d's figure (f32 shard once per batch), NOT a memory rate
                          "streamed_gbs": streamed / (t / args.steps) / 1e9, "streamed_bytes_per_batch": streamed})
            if p == "exact" and k_n and nq in (1, 16):
                ach = rows * dim * 4 / (k_ms / k_n * 1e-3) / 1e9
                cand = {"bound": "hbm", "kernel": "scan_exact_kernel", "queries": nq, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": rows * dim * 4, "avg_launch_ms": k_ms / k_n, "launches": k_n,
                        "end_to_end_gbs": rows * dim * 4 / (t / args.steps) / 1e9,
                        "traffic": ((pmc_traffic() or {}).get("scan_exact_kernel", {}).get("fetch_bytes_x2_per_launch") or None) if (rows, dim) == (10_000_000, 768) else None,
                        "traffic_note": f"HBM-side bytes per launch, {(pmc_traffic() or {}).get('_file')} (one corpus pass; f32 shard = rows*dim*4)"}
                if exact_roof is None or nq == 1:
                    exact_roof = cand
        ix.set_search_mode("auto")
    ix.close()

    # The same batch on CLUSTERED corpora (4 096 Gaussian clusters: what code embeddings look like more than i.i.d. rows do).
    # The 512-candidate certificate of the int8 stage does not hold there (more than 512 rows lie within the int8 error of the k-th
    # neighbour): the collect pass answers the first batch, and from the second on the stage runs its wide form -- thresholds from
    # exact scores before every phase, every key within the cut kept (up to 4 096 per query) -- which answers in one pass.
    # Reported: the steady-state batch (after three searches).
    clustered = None
    if not args.no_sweep and world == 1:
        clustered = {}
        for tag, spread in (("spread_0.5", 0.5), ("spread_0.1", 0.1)):
            cx = _native.Index(rt, dim, metric=args.metric_type, kind="FLAT")
            cx.fill_synthetic_clustered(rows, seed=0, nclusters=4096, spread=spread)
            qs = _native.Index(rt, dim, metric=args.metric_type)
            qs.fill_synthetic_clustered(Q, seed=0, nclusters=4096, spread=spread, first_row=rows + 12345)
            qc = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
            qs.close()
            cx.search_dev(qc.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr())
            first = cx.last_search_stats()
            for _ in range(2):
                cx.search_dev(qc.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr())
            nst = max(2, args.steps // 2)
            tc = timed(ctx, lambda: cx.search_dev(qc.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr()), nst)
            stc = cx.last_search_stats()
            brief = lambda st: {"first_stage": ("int8" if st.get("coarse_bits", 16) == 8 else "bf16") + (" (wide candidate set)" if st.get("wide") else ""),
                                "collect_pass_resolved_of_tried": [st.get("collect_resolved", 0), st.get("collect_tried", 0)],
                                "handed_to_bf16": st.get("handed_to_bf16", 0), "uncertified": st["uncertified"]}
            clustered[tag] = {"workload": f"{rows} x {dim} f32 rows in 4096 Gaussian clusters (spread {spread}), batch-{Q} queries drawn from the same clusters, {args.metric_type} top-{k}",
                              "ms_per_step": 1e3 * tc / nst, "value": Q * nst / tc, "unit": "queries/s", **brief(stc), "first_batch": brief(first)}
            cx.close()
        clustered["note"] = ("wide candidate set = the int8 stage keeps every key within k-th exact score + coarse error bound between its phases: certified by construction; "
                             "collect pass = a second pass at the same precision with that threshold fixed (what answers the first batch, and queries whose set exceeds 4 096 keys); "
                             "in round 2 spread 0.1 sent every query to the exact scan: 347 ms per batch (profiles/r3t_clustered_probe.log)")

    stats = None
    alg_bytes = rows * dim * 4  # the f32 shard is read once per query batch (+ norms, negligible)
    ld = (dim + 63) // 64 * 64
    step_s = dt / args.steps
    if path == "batched":
        # dominant kernel = scan_coarse256_kernel (MFMA GEMM of the coarse shadow against the query batch + threshold filter), several
        # phase launches per step; int8 stage: v_mfma_i32_16x16x64_i8 on the int8 shadow (peak 2x the bf16 rate), else bf16
        i8 = coarse_bits == 8
        qpad = (Q + 255) // 256 * 256 if (i8 or Q > 128) else 128
        flops = 2.0 * rows * ld * qpad * args.steps
        achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_n else None
        peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_BF16_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": "scan_coarse256_kernel<int8>" if i8 else "scan_coarse256_kernel<bf16>", "achieved": achieved, "peak": peak,
                "unit": "TOP/s" if i8 else "TFLOP/s", "frac": achieved / peak if achieved else None,
                "frac_of_bf16_peak": achieved / MFMA_BF16_PEAK_TFLOPS if achieved else None,
                "traffic": coarse_traffic(rows, Q) if dim == 768 else None,
                "traffic_note": f"HBM-side bytes per step over the phase launches (coarse shadow = rows*ld*{1 if i8 else 2} B), {(pmc_traffic() or {}).get('_file')}",
                "coarse_stage": "int8 (per-row scale, 512 candidates/query)" if i8 else "bf16 (128 candidates/query)", "queries_handed_to_bf16_stage": handed,
                "algorithmic_flops_per_step": flops / args.steps, "kernel_ms_per_step": scan_ms / args.steps, "launches": scan_n,
                "select_ms_per_step": merge_ms / args.steps,
                "hbm_view": {"algorithmic_bytes_per_step": alg_bytes, "achieved_gbs": alg_bytes / step_s / 1e9,
                             "frac_of_8TBs": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                             "note": "algorithmic bytes = the f32 shard once per batch (SURVEY 8d); at Q=1024 the binding roof is the matrix pipe: "
                                     "2*rows*dim*Q = 1.57e13 ops = 3.1 ms at the int8 peak / 6.3 ms at the bf16 peak, vs 1.0 / 1.9 ms to stream the 7.7 / 15.4 GB shadow"}}
    else:
        kern_ms = scan_ms / max(1, scan_n)
        achieved = alg_bytes * groups / (kern_ms * 1e-3) / 1e9 if scan_n else None
        roof = {"bound": "hbm", "kernel": "scan_exact_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": ((pmc_traffic() or {}).get("scan_exact_kernel", {}).get("fetch_bytes_x2_per_launch", 0) * groups or None) if (rows, dim) == (10_000_000, 768) else None,
                "algorithmic_bytes_per_launch": alg_bytes * groups, "avg_launch_ms": kern_ms, "launches": scan_n,
                "merge_avg_ms": merge_ms / max(1, merge_n)}
    res = {
        "value": Q * args.steps / dt,
        "unit": "queries/s",
        "ms_per_step": 1e3 * step_s,
        "rows_scanned_per_s": Q * args.steps / dt * rows * world,
        "workload": f"brute-force {args.metric_type} top-{k}, {rows} x {dim} f32 rows per GPU ({rows * world} total), batch-{Q} queries",
        "dtype": (("int8" if coarse_bits == 8 else "bf16") + " coarse + f32 exact re-rank") if path == "batched" else "f32",
        "path": path, "uncertified_queries_last_step": unc,
        "roofline": roof,
    }
    if world > 1:
        res["host_merge_ms_per_step"] = (1e3 * merge_time[0] / merge_time[1]) if merge_time[1] else None  # rank 0, overlapped with the next step's scan
    if merger:
        merger.shutdown()
    ctx.drain = None
    if clustered:
        res["clustered"] = clustered
    if sweep:
        res["sweep"] = sweep
    if exact_roof:
        res["exact_scan_roofline"] = exact_roof  # the HBM-bound regime of the same path (Q <= 16): north_star's ">= 60 % of HBM roofline"
    return res


def coarse_traffic(rows: int, Q: int):
    """Measured fetch (+ write) of the coarse kernel's phase launches per step at 10M x 768, Q = 1024."""
    t = pmc_traffic()
    if not t or (rows, Q) != (10_000_000, 1024):
        return None
    c = t.get("scan_coarse256_kernel", {})
    if "fetch_bytes_x2_per_step" in c:
        return c["fetch_bytes_x2_per_step"] + (c.get("write_bytes_per_step") or 0.0)
    keys = ["grid_35264512", "grid_33554432", "grid_2097152", "grid_524288", "grid_131072", "grid_32768", "grid_8192"]  # r1k layout
    total = sum(c.get(k, {}).get("fetch_bytes_x2_median", 0) for k in keys)
    return total + 1_048_576 * 768 * 2 * 1.09 if total else None


def cpu_baseline_scan(args) -> dict:
    """CPU twins of the scan on a bounded sample (all host cores, count stated): the value is the sgemm formulation a CPU
    implementation would use (|x|^2 + |q|^2 - 2 X Q^T by BLAS + partial sort: oracle.sc_oracle.search_sgemm); the canonical
    fmaf-chain C port that defines the bit-exact arithmetic (the parity oracle, OpenMP over rows) is timed beside it."""
    from oracle import sc_oracle as orc

    cores = orc.host_cores()  # affinity mask capped by the cgroup CPU quota
    rows, nq = 1_000_000, args.queries
    X = orc.synth(rows, args.dim, seed=0)
    Q = orc.synth(nq, args.dim, seed=1)
    orc.search_sgemm(X[:20_000], Q[:64], args.k, args.metric_type, threads=cores)
    t0 = time.perf_counter()
    sd, sr = orc.search_sgemm(X, Q, args.k, args.metric_type, threads=cores)
    dt = time.perf_counter() - t0
    crow = 200_000
    orc.search(X[:2000], Q[:4], args.k, args.metric_type)  # warm the thread pool
    t0 = time.perf_counter()
    cd, cr = orc.search(X[:crow], Q, args.k, args.metric_type)
    dtc = time.perf_counter() - t0
    # BASELINE.md section 3, as written: numpy |x|^2 - 2 Q X^T (sgemm) + argpartition + stable argsort over 100 000 x 768 (default_rng(0)),
    # 1 024 queries (default_rng(1)), k = 10, L2; 1 warm-up, median of 5, unscaled
    import numpy as np

    Xp = np.random.default_rng(0).standard_normal((100_000, 768), dtype=np.float32)
    Qp = np.random.default_rng(1).standard_normal((1024, 768), dtype=np.float32)

    def protocol_once():
        t0 = time.perf_counter()
        d2 = (Xp * Xp).sum(1)[None, :] - 2.0 * (Qp @ Xp.T)
        part = np.argpartition(d2, 10, axis=1)[:, :10]
        pd = np.take_along_axis(d2, part, axis=1)
        order = np.lexsort((part, pd), axis=1)  # distance, then the lower row
        np.take_along_axis(part, order, axis=1)
        return time.perf_counter() - t0

    protocol_once()
    times = sorted(protocol_once() for _ in range(5))
    tp = times[2]
    protocol = {"value": 1024 / tp, "unit": "queries/s", "cores": cores, "seconds_per_batch": tp, "algorithmic_gbs": 100_000 * 768 * 4 / tp / 1e9,
                "sample": "BASELINE.md section 3 protocol: numpy sgemm + argpartition + stable sort, 100 000 x 768 f32 (default_rng(0)), 1 024 queries (default_rng(1)), "
                          "k = 10, L2; 1 warm-up, median of 5, unscaled (a 100x smaller corpus than the GPU leg's)"}
    return {"value": nq / (dt * (args.rows / rows)), "unit": "queries/s", "cores": cores, "kind": "port",
            "baseline_md_protocol": protocol,
            "sample": f"sgemm + partial sort (torch CPU, f32) over {rows} of {args.rows} rows x all {nq} queries in {dt:.2f}s, scaled linearly in rows (exhaustive scan)",
            "canonical_port": {"value": nq / (dtc * (args.rows / crow)), "unit": "queries/s", "cores": orc.threads(),
                               "sample": f"scalar fmaf-chain C port (the parity oracle), {crow} rows x {nq} queries in {dtc:.2f}s, scaled linearly"}}


# -------------------------------------------------------------------------------------------- ivf

def bench_ivf(ctx, args) -> dict:
    """BASELINE configs[4]: IVF_FLAT nlist 4096 / nprobe 64 over 10M x 3072, batch-1024 queries, k 10, L2 (index parameters of
    reference milvus_store.py:76-83,141-147 at benchmark scale).  Corpus: clustered synthetic rows generated on device (one
    true cluster per list); queries: rows of the same distribution that are not in the corpus.  Ground truth for recall@10 =
    the exhaustive (certified-exact) search of the same index.  One step = one list-major probe of the whole batch."""
    import numpy as np
    import torch

    from semcode_amd import _native

    rows, dim, nlist, nprobe, Q, k = args.ivf_rows, args.ivf_dim, args.ivf_nlist, args.ivf_nprobe, args.queries, args.k
    rt, dev = ctx.rt, ctx.dev
    ix = _native.Index(rt, dim, metric="L2", kind="IVF_FLAT", nlist=nlist)
    ix.fill_synthetic_clustered(rows, seed=0, nclusters=nlist, spread=0.5)
    qsrc = _native.Index(rt, dim, metric="L2")
    qsrc.fill_synthetic_clustered(Q, seed=0, nclusters=nlist, spread=0.5, first_row=rows + 12345)
    qh = qsrc.get_rows(0, Q)
    qsrc.close()
    q = torch.from_numpy(qh).to(dev)
    out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
    out_r = torch.empty((Q, k), dtype=torch.int64, device=dev)
    ix.search_dev(q.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr(), nprobe=nlist)  # untrained: exhaustive
    rt.synchronize()
    t0 = time.perf_counter()
    ix.search_dev(q.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr(), nprobe=nlist)
    rt.synchronize()
    t_bf = time.perf_counter() - t0
    bf_stats = ix.last_search_stats()
    truth = out_r.cpu().numpy().copy()
    ix.release_scratch()  # the 61 GB bf16 shadow makes room for the second corpus copy of the build
    t0 = time.perf_counter()
    ix.train(niter=10)
    rt.synchronize()
    t_train = time.perf_counter() - t0
    sizes = ix.ivf_info()["list_sizes"]

    def step():
        ix.search_dev(q.data_ptr(), Q, k, out_d.data_ptr(), out_r.data_ptr(), nprobe=nprobe)

    def leg(mode):
        ix.set_search_mode(mode)
        for _ in range(max(1, args.warmup)):
            step()
        rt.set_profiling(True)
        rt.profile_reset()
        dt = timed(ctx, step, args.steps)
        k_ms, k_n = rt.profile_read(0)
        m_ms, _ = rt.profile_read(1)
        rt.set_profiling(False)
        st = ix.last_search_stats()
        return {"dt": dt, "k_ms": k_ms, "k_n": k_n, "m_ms": m_ms, "path": st["path"], "uncertified": st["uncertified"], "pst": ix.last_probe_stats(),
                "ids": out_r.cpu().numpy().copy(), "dist": out_d.cpu().numpy().copy()}

    # the product's own choice ("auto": list-major probing behind the int8 coarse stage for such a batch) is the timed step;
    # the exact f32 list-major probe is timed beside it and must return the same bits
    auto = leg("auto")
    exact = leg("ivf_listmajor")
    same = bool(np.array_equal(auto["ids"], exact["ids"]) and np.array_equal(auto["dist"].view(np.uint32), exact["dist"].view(np.uint32)))
    recall = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(auto["ids"].tolist(), truth.tolist())]))
    ix.set_search_mode("auto")  # one query: the planner's own choice (the coarse stage when the probe streams a gigabyte or more)
    for _ in range(2):
        ix.search_dev(q.data_ptr(), 1, k, out_d.data_ptr(), out_r.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ix.search_dev(q.data_ptr(), 1, k, out_d.data_ptr(), out_r.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    t_one = (time.perf_counter() - t0) / 3
    one_path = ix.last_search_stats()["path"]
    ix.close()
    step_s = auto["dt"] / args.steps
    ld = (dim + 63) // 64 * 64
    ld8 = (ld + 127) // 128 * 128
    pst = auto["pst"]
    coarse = auto["path"] == "ivf_coarse"
    row_bytes = ld8 if coarse else ld * 4  # what the scan kernels of the timed path read per row
    streamed, unique = pst["streamed_rows"] * row_bytes, pst["unique_rows"] * row_bytes
    unique_f32 = pst["unique_rows"] * ld * 4
    kern_s = (auto["k_ms"] / args.steps) * 1e-3 if auto["k_n"] else None
    ex_s = exact["dt"] / args.steps
    ex_kern_s = (exact["k_ms"] / args.steps) * 1e-3 if exact["k_n"] else None
    ex_unique = exact["pst"]["unique_rows"] * ld * 4
    return {"value": Q / step_s, "unit": "queries/s", "ms_per_step": 1e3 * step_s, "recall_at_10": recall, "path": auto["path"],
            "uncertified": auto["uncertified"], "same_bits_as_exact_listmajor": same,
            "workload": f"IVF_FLAT nlist={nlist} nprobe={nprobe}, {rows} x {dim} f32 rows (clustered synthetic, device generated), batch-{Q} queries, L2 top-{k}",
            "train_s": t_train, "list_size_min_median_max": [int(sizes.min()), int(np.median(sizes)), int(sizes.max())],
            "roofline": {"bound": "hbm", "kernel": "scan_coarse64s_kernel<L2, GROUPED> (int8 lists centred on their centroids, 64 (query, list) slots per group; phase A dense, phase B against the exact bound) + the quantizer's scan_exact_kernel"
                                   if coarse else "scan_listgemm_kernel + scan_exact_kernel (segment mode)",
                         "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": unique / kern_s / 1e9 if kern_s else None, "frac": unique / kern_s / 1e9 / HBM_PEAK_GBS if kern_s else None,
                         "algorithmic_bytes_per_step": unique, "streamed_bytes_per_step": streamed, "groups": pst["groups"],
                         "kernel_ms_per_step": auto["k_ms"] / args.steps if auto["k_n"] else None,
                         "select_rerank_ms_per_step": auto["m_ms"] / args.steps,
                         "algorithmic_f32_gbs": unique_f32 / step_s / 1e9,
                         "note": "algorithmic = bytes of the DISTINCT probed lists in the form the timed path streams them (int8 shadow: 1 B per padded element; "
                                 "SURVEY 8d's f32 figure is algorithmic_f32_gbs, whole step, NOT a memory rate); streamed = what the scan kernels read "
                                 "(a list is streamed once per group of up to 64 queries that probe it; re-reads mostly hit L2 / MALL)"},
            "exact_listmajor": {"ms_per_batch": 1e3 * ex_s, "qps": Q / ex_s, "path": exact["path"], "kernel_ms_per_step": exact["k_ms"] / args.steps if exact["k_n"] else None,
                                "hbm_frac_f32_lists": ex_unique / ex_kern_s / 1e9 / HBM_PEAK_GBS if ex_kern_s else None,
                                "f32_mfma_view": {"algorithmic_flops_per_step": 2.0 * Q * nprobe * (rows / nlist) * dim, "peak_tflops": 157.3,
                                                  "achieved_tflops": 2.0 * Q * nprobe * (rows / nlist) * dim / ex_kern_s / 1e12 if ex_kern_s else None,
                                                  "note": "exact f32 scores of every (query, probed row) pair on v_mfma_f32_16x16x4_f32"}},
            "exhaustive": {"ms_per_batch": 1e3 * t_bf, "qps": Q / t_bf, "path": bf_stats["path"], "uncertified": bf_stats["uncertified"]},
            "single_query_ms": 1e3 * t_one, "single_query_path": one_path}


# ------------------------------------------------------------------------------------------- main

def main() -> None:
    args = parse()
    import torch

    from semcode_amd import _native

    ctx = Ctx()
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ctx.world == 1 and args.gpus > 1:
        raise SystemExit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
    # one process per GPU; SEMCODE_BENCH_BACKEND=gloo lets several ranks share one device (rehearsal on a 1-GPU box)
    backend = os.environ.get("SEMCODE_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    ctx.dev = torch.device("cuda", local)
    ctx.dist = None
    ctx.backend = backend
    ctx.comm = None
    ctx.nccl_group = None
    ctx.collective = None
    stream = torch.cuda.Stream(device=ctx.dev)
    ctx.rt = _native.Runtime(device=local, stream=stream.cuda_stream)
    info = ctx.rt.device_info()
    if ctx.world > 1:
        import torch.distributed as dist

        # control plane: a gloo group started from the launcher's MASTER_ADDR / MASTER_PORT.  It carries the 128-byte RCCL
        # rendezvous id (and, for the rehearsal backend, the exchange itself).  The data path's collective is the native
        # communicator: RCCL bound by libsemcode_hip, issued on the runtime's stream (include/semcode_hip.h, sc_comm_*).
        dist.init_process_group("gloo")
        ctx.dist = dist
        if backend == "nccl":
            from semcode_amd.storage.sharded import make_comm

            err = ""
            try:
                ctx.comm = make_comm(ctx.rt)
                ctx.comm.allreduce_max(0.0)
            except Exception as exc:  # every rank must take the same path: agree on the outcome first
                err = f"{type(exc).__name__}: {exc}"
                ctx.comm = None
            ok = torch.tensor([0 if err else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                ctx.collective = "RCCL all-gather behind the C ABI (sc_index_search_sharded_dev), one grouped launch per step"
            else:
                # stated loudly in the output line: the exchange then goes through torch.distributed's RCCL binding instead
                if ctx.comm is not None:
                    ctx.comm.close()
                    ctx.comm = None
                errs = [None] * ctx.world
                dist.all_gather_object(errs, err)
                ctx.nccl_group = dist.new_group(backend="nccl", device_id=ctx.dev)
                ctx.collective = "torch.distributed nccl all_gather_into_tensor (native sc_comm unavailable: " + "; ".join(e for e in errs if e)[:300] + ")"
                print("bench.py: native communicator failed, using torch.distributed nccl:", errs, file=sys.stderr, flush=True)
        else:
            ctx.collective = f"torch.distributed {backend} on the host (rehearsal: several ranks on one device)"

    embed = scan = ivf = None
    with torch.cuda.stream(stream):
        if args.workload in ("both", "embed"):
            embed = bench_embed(ctx, args)
        if args.workload in ("both", "scan"):
            scan = bench_scan(ctx, args)
        if args.workload == "both" and ctx.world == 1 and not args.no_ivf:
            ivf = bench_ivf(ctx, args)

    line = {
        "metric": "chunks/sec embed (256-tok) + top-k QPS over 10M x 768",
        "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "data": "synthetic",
    }
    if ctx.collective:
        line["collective"] = ctx.collective
    if ctx.comm is not None:
        ci = ctx.comm.info()  # what the native communicator itself reports: proof that RCCL saw N ranks
        line["rccl_ranks"], line["rccl_version"] = ci["world"], ci["rccl_version"]
    if embed:
        line.update({"value": embed["chunks_per_s"], "unit": "chunks/s", "ms_per_step": embed["ms_per_step"], "dtype": "bf16",
                     "config": {"workload": embed["workload"], "global_batch": args.batch * ctx.world, "seq_len": args.seq,
                                "parallelism": f"dp{ctx.world} (replicated weights, no collective)", "device": info["name"]},
                     "roofline": embed["roofline"]})
        if scan:
            line["topk"] = scan
        if ivf:
            line["ivf"] = ivf
    else:
        line.update({"value": scan["value"], "unit": "queries/s", "ms_per_step": scan["ms_per_step"], "dtype": "f32",
                     "config": {"workload": scan["workload"], "sharding": f"row-range x{ctx.world}", "device": info["name"]},
                     "roofline": scan["roofline"]})
        if "sweep" in scan:
            line["sweep"] = scan["sweep"]
    if ctx.rank == 0:
        if ctx.world == 1 and not args.no_cpu_baseline:
            if embed:
                line["cpu_baseline"] = cpu_baseline_embed(args)
                if scan:
                    line["topk"]["cpu_baseline"] = cpu_baseline_scan(args)
            else:
                line["cpu_baseline"] = cpu_baseline_scan(args)
        print(json.dumps(line), flush=True)
    if ctx.comm is not None:
        ctx.comm.close()
    ctx.rt.close()
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
