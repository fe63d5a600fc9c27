"""How fast does the exact kernel stream in its three uses on one IVF index?  (tuning aid)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from semcode_amd import _native

dim = 3072
nlist = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
quick = len(sys.argv) > 3  # list-major only
rt = _native.Runtime(0)
ix = _native.Index(rt, dim, metric="L2", kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=nlist, spread=0.5)
qsrc = _native.Index(rt, dim, metric="L2")
qsrc.fill_synthetic_clustered(1024, seed=0, nclusters=nlist, spread=0.5, first_row=rows + 12345)
Q = qsrc.get_rows(0, 1024)
qsrc.close()
ix.train(niter=6)
gb = rows * dim * 4 / 1e9


def timed(fn, reps=3):
    fn()
    t = time.time()
    for _ in range(reps):
        fn()
    return (time.time() - t) / reps


ix.set_search_mode("exact")
for q in (() if quick else (1, 6, 12)):
    dt = timed(lambda: ix.search(Q[:q], k=10))
    print(f"exhaustive exact, Q={q:3d}: {dt * 1e3:8.2f} ms  ({-(-q // 6)} passes of {gb:.1f} GB -> {-(-q // 6) * gb / dt / 1e3:.2f} TB/s)", flush=True)
ix.set_search_mode("ivf")
for q, nprobe in (() if quick else ((1, nlist // 16), (1, nlist // 4), (8, nlist // 16))):
    dt = timed(lambda: ix.search(Q[:q], k=10, nprobe=nprobe))
    print(f"per-query probing, Q={q}, nprobe={nprobe}: {dt * 1e3:8.2f} ms  ({q * nprobe / nlist * gb / dt / 1e3:.2f} TB/s)", flush=True)
ix.set_search_mode("ivf_listmajor")
for q, nprobe in ((256, nlist // 64), (1024, nlist // 64)) + (() if quick else ((1024, nlist // 16),)):
    dt = timed(lambda: ix.search(Q[:q], k=10, nprobe=nprobe))
    print(f"list-major, Q={q}, nprobe={nprobe}: {dt * 1e3:8.2f} ms", flush=True)
ix.close()
rt.close()
