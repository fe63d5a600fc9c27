"""Batched scan over the Gaussian 10M x 768 corpus: the int8 stage in its plain form (512 candidates + certificate) against the wide
form (every key within the exact-score cut), same index, same batch."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

rows, dim, Q, k = 10_000_000, 768, 1024, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="L2")
ix.fill_synthetic(rows, seed=0)
qs = _native.Index(rt, dim, metric="L2")
qs.fill_synthetic(Q, seed=1)
q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
qs.close()
od = torch.empty((Q, k), dtype=torch.float32, device=dev)
orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
res = {}
for wide in (0, 1, 0, 1):
    _native.diag_set_option("wide_candidates", wide)
    for _ in range(3):
        ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
    rt.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
    rt.synchronize()
    dt = (time.perf_counter() - t0) / 10
    st = ix.last_search_stats()
    res[wide] = (orow.cpu().numpy().copy(), od.cpu().numpy().copy())
    print(f"wide {wide}: {dt * 1e3:7.2f} ms / batch   {st}", flush=True)
print("same ids:", np.array_equal(res[0][0], res[1][0]), " same bits:", np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)))
_native.diag_set_option("wide_candidates", 0)
ix.close(); rt.close()
