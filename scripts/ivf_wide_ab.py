"""A/B of list-major IVF probing with and without the 64-query groups (SC_IVF_WIDE), one process, one index.

    python scripts/ivf_wide_ab.py [--rows 2000000 --dim 3072 --nlist 1024 --nprobe 64 --queries 1024]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from semcode_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--dim", type=int, default=3072)
ap.add_argument("--nlist", type=int, default=1024)
ap.add_argument("--nprobe", type=int, default=64)
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
rt = _native.Runtime(0)
ix = _native.Index(rt, a.dim, metric="L2", kind="IVF_FLAT", nlist=a.nlist)
ix.fill_synthetic_clustered(a.rows, seed=0, nclusters=a.nlist, spread=0.5)
qsrc = _native.Index(rt, a.dim, metric="L2")
qsrc.fill_synthetic_clustered(a.queries, seed=0, nclusters=a.nlist, spread=0.5, first_row=a.rows + 12345)
Q = qsrc.get_rows(0, a.queries)
qsrc.close()
t = time.time(); ix.train(niter=4); print("train %.2f s" % (time.time() - t), flush=True)
ix.set_search_mode("ivf_listmajor")
res = {}
for wide in ("1", "0", "1", "0"):
    os.environ["SC_IVF_WIDE"] = wide
    d, r = ix.search(Q, k=a.k, nprobe=a.nprobe)
    ts = []
    for _ in range(a.reps):
        t = time.time(); d, r = ix.search(Q, k=a.k, nprobe=a.nprobe); ts.append(time.time() - t)
    st = ix.last_probe_stats() if hasattr(ix, "last_probe_stats") else {}
    print(f"wide={wide}: {1e3 * min(ts):.2f} ms best, {1e3 * np.median(ts):.2f} median  {st}", flush=True)
    res.setdefault(wide, (d.copy(), r.copy()))
assert np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][0].view(np.uint32), res["0"][0].view(np.uint32))
print("identical results")
ix.close(); rt.close()
