"""Exhaustive exact search latency against k (tuning aid): the merge of the per-workgroup lists grows with k."""
import sys
import time

sys.path.insert(0, ".")
from semcode_amd import _native

rt = _native.Runtime(0)
ix = _native.Index(rt, 768, metric="L2")
ix.fill_synthetic(2_000_000, seed=1)
qs = _native.Index(rt, 768, metric="L2")
qs.fill_synthetic(4, seed=2)
Q = qs.get_rows(0, 4)
ix.set_search_mode("exact")
for k in (10, 32, 33, 64, 100, 256, 1024):
    ix.search(Q, k=k)
    t = time.time()
    for _ in range(10):
        ix.search(Q, k=k)
    print(f"k={k:5d}: {(time.time() - t) / 10 * 1e3:8.3f} ms per search (2M x 768, 4 queries)", flush=True)
