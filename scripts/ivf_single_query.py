"""Single-query IVF latency at 3 072 dimensions (tuning aid): where do the ~2 ms go?   python3 scripts/ivf_single_query.py [rows nlist nprobe]"""
import sys
import time

sys.path.insert(0, ".")
from semcode_amd import _native

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dim = 3072
rt = _native.Runtime(0)
ix = _native.Index(rt, dim, metric="L2", kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=nlist, spread=0.5)
qsrc = _native.Index(rt, dim, metric="L2")
qsrc.fill_synthetic_clustered(8, seed=0, nclusters=nlist, spread=0.5, first_row=rows + 12345)
Q = qsrc.get_rows(0, 8)
ix.train(niter=4)
sizes = ix.ivf_info()["list_sizes"]
for nq in (1, 4):
    ix.search(Q[:nq], k=10, nprobe=nprobe)
    t = time.time()
    for _ in range(20):
        ix.search(Q[:nq], k=10, nprobe=nprobe)
    ms = (time.time() - t) / 20 * 1e3
    print(f"rows={rows} nlist={nlist} nprobe={nprobe} Q={nq}: {ms:.3f} ms per search, path {ix.last_search_stats()['path']}, "
          f"~{nprobe * sizes.mean() * dim * 4 / 1e9:.2f} GB probed per query", flush=True)
