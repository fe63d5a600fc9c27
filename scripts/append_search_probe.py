"""Appends of 128 rows alternating with searches of 32 queries on a FLAT index (10M x 768, capacity reserved): the int8 shadow follows the
appended rows alone -- no step of the search cost (profiles/r3z_overwrite_search_interleave.log, last block)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from semcode_amd import _native
rows, dim, Q, k = 10_000_000, 768, 32, 5
stream = torch.cuda.Stream(); rt = _native.Runtime(device=0, stream=stream.cuda_stream); dev = torch.device("cuda", 0)
fx = _native.Index(rt, dim, metric="IP")
fx.reserve(rows + 100_000)
fx.fill_synthetic_clustered(rows, seed=0, nclusters=1024, spread=0.5)
q = torch.empty((Q, dim), dtype=torch.float32, device=dev); rt.synth_fill_dev(q.data_ptr(), Q, dim, dim, seed=5)
new = np.random.default_rng(0).standard_normal((128, dim)).astype(np.float32)
od = torch.empty((Q, k), dtype=torch.float32, device=dev); orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
def fsearch():
    rt.synchronize(); t0 = time.perf_counter()
    fx.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr()); rt.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(3): fsearch()
print(f"steady {fsearch():.2f} ms {fx.last_search_stats()}", flush=True)
for it in range(8):
    t0 = time.perf_counter(); fx.add(new); rt.synchronize(); ta = (time.perf_counter() - t0) * 1e3
    t1 = fsearch(); st = fx.last_search_stats(); t2 = fsearch()
    print(f"append 128: {ta:.2f} ms -> first search {t1:.2f} ms, second {t2:.2f} ms {st}", flush=True)
