"""Differential sweep over the search planner: random (rows, dim, metric, index kind, batch, top_k, nprobe, corpus shape) and every
path the planner can take for them, each compared bit for bit (ids and f32 distances) with the exact path of the same index
(FLAT: the exact f32 scan; IVF_FLAT: per-query probing).  Prints one line per case; exits 1 on the first difference.

    python scripts/planner_fuzz.py [seed] [cases]
"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

def run(rt, dev, ix, q, k, nprobe, mode):
    Q = q.shape[0]
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
    ix.set_search_mode(mode)
    ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    return orow.cpu().numpy(), od.cpu().numpy(), ix.last_search_stats()


def sweep(seed: int, cases: int) -> int:
    """-> the number of (case, path) pairs whose results differ from their reference."""
    rng = np.random.default_rng(seed)
    stream = torch.cuda.Stream()
    rt = _native.Runtime(device=0, stream=stream.cuda_stream)
    dev = torch.device("cuda", 0)
    bad = 0
    t_all = time.time()
    for case in range(cases):
        rows = int(rng.choice([3000, 70_000, 130_000, 300_000, 1_100_000, 2_500_000]))
        dim = int(rng.choice([64, 128, 384, 768, 1024, 7, 100, 200, 1000, 1536, 3072]))
        if rows * dim > 1_500_000_000:
            dim = 384
        metric = str(rng.choice(["L2", "IP", "COSINE"]))
        kind = str(rng.choice(["FLAT", "IVF_FLAT"]))
        Q = int(rng.choice([1, 3, 16, 17, 40, 64, 65, 128, 129, 300, 1024]))
        k = int(rng.choice([1, 5, 10, 64, 65, 100, 128, 129, 200]))
        shape = str(rng.choice(["uniform", "clustered", "tight"]))
        nlist = int(rng.choice([16, 128, 1024])) if kind == "IVF_FLAT" else 0
        if nlist * 8 > rows:
            nlist = 16
        nprobe = int(rng.choice([1, 4, 16, 64])) if kind == "IVF_FLAT" else 16
        nprobe = min(nprobe, max(1, nlist))
        ix = _native.Index(rt, dim, metric=metric, kind=kind, nlist=max(nlist, 1))
        if shape == "uniform":
            ix.fill_synthetic(rows, seed=case)
        else:
            ix.fill_synthetic_clustered(rows, seed=case, nclusters=int(rng.choice([8, 200, 3000])), spread=0.02 if shape == "tight" else 0.3)
        q = torch.empty((Q, dim), dtype=torch.float32, device=dev)
        if dim % 4 == 0:
            rt.synth_fill_dev(q.data_ptr(), Q, dim, dim, seed=1000 + case)
        else:
            q.copy_(torch.from_numpy(rng.uniform(-1, 1, size=(Q, dim)).astype(np.float32)))
        # some queries are corpus rows (distance 0 / a tie with a planted duplicate), one query is repeated
        nself = min(Q, 1 + Q // 8)
        pick = rng.integers(0, rows, size=nself)
        for j, r in enumerate(pick):
            q[j].copy_(torch.from_numpy(ix.get_rows(int(r), 1)[0]).to(dev))
        if Q > 2:
            q[Q - 1].copy_(q[0])
        torch.cuda.synchronize()
        events = []
        if kind == "IVF_FLAT":
            ix.train(niter=4, seed=case)
            events.append("trained")
            if rng.random() < 0.5:  # rows appended after training: the tail segment / the re-layout
                extra = int(rng.choice([7, 900, 70_000]))
                ix.add(ix.get_rows(0, min(extra, rows)))
                events.append(f"+{min(extra, rows)}")
        elif rng.random() < 0.3:
            extra = int(rng.choice([5, 3000]))
            ix.add(ix.get_rows(0, extra))  # duplicates of rows 0..extra-1: ties by row id
            events.append(f"+{extra}dup")
        # the reference of a path: the per-query probe when the answer came from the lists (asked right AFTER the path ran, so both see
        # the same lists and the same tail -- an exhaustive search in between folds the tail into the lists), the exact f32 scan when
        # the planner answered exhaustively (small corpora, single queries with a large top_k: exact results, not the probe's)
        modes = ["auto", "ivf_listmajor", "ivf_coarse", "ivf", "auto"] if kind == "IVF_FLAT" else ["auto", "batched", "auto"]
        line = []
        for rnd in range(2):
            if rnd == 1:
                # second round on the SAME index after it changed under its shadows (int8 / bf16 / centred copies, norms, lists):
                # rows overwritten with the queries themselves (each must now be its query's nearest row), or rows appended
                what = str(rng.choice(["overwrite", "append", "both"]))
                qh = q.cpu().numpy()
                if what in ("overwrite", "both"):
                    m = min(Q, 50)
                    tgt = rng.choice(len(ix), size=m, replace=False).astype(np.int64)
                    ix.overwrite(qh[:m], tgt)
                if what in ("append", "both"):
                    ix.add(qh[: min(Q, 30)] * np.float32(0.5))
                line.append(f"[{what}]")
            for mode in modes:
                try:
                    r, d, st = run(rt, dev, ix, q, k, nprobe, mode)
                except _native.ScError as e:
                    line.append(f"{mode}: refused ({e.args[0] if e.args else e})")
                    continue
                ref_mode = "exact" if st["path"] in ("exact", "batched") else "ivf"
                ref_rows, ref_d, rst = run(rt, dev, ix, q, k, nprobe, ref_mode)
                same = np.array_equal(r, ref_rows) and np.array_equal(d.view(np.uint32), ref_d.view(np.uint32))
                if bool(st.get("tail_rows")) != bool(rst.get("tail_rows")):
                    line.append(f"{mode}->{st['path']}: tail state differs from the reference's (not compared)")
                    continue
                line.append(f"{mode}->{st['path']}{'/wide' if st.get('wide') else ''}{'/tail' if st.get('tail_rows') else ''} vs {rst['path']}: {'same' if same else 'DIFFERENT'}")
                if rnd == 1 and metric == "L2" and shape != "tight" and what != "append" and not all(np.array_equal(ix.get_rows(int(r[j, 0]), 1)[0], qh[j]) for j in range(m)):
                    same = False  # (an overwritten row IS its query: distance 0 on every path, whatever the reference says)
                    line.append("an overwritten row was not found")
                if not same:
                    bad += 1
                    diffq = np.nonzero((r != ref_rows).any(axis=1) | (d.view(np.uint32) != ref_d.view(np.uint32)).any(axis=1))[0]
                    j = int(diffq[0]) if len(diffq) else 0
                    print(f"  first differing query {j} of {len(diffq)}: stats {st}\n   ref rows {ref_rows[j][:12]} d {ref_d[j][:6]}\n   got rows {r[j][:12]} d {d[j][:6]}", flush=True)
        print(f"case {case}: rows {rows} dim {dim} {metric} {kind} nlist {nlist} nprobe {nprobe} Q {Q} k {k} {shape} {' '.join(events)} | " + " | ".join(line), flush=True)
        ix.close()
        del q
        torch.cuda.empty_cache()
    print(f"{cases} cases, {bad} differences, {time.time() - t_all:.0f} s", flush=True)
    rt.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60) else 0)
