"""Clustered 10M x 768 corpus (4 096 clusters, spread 0.1): small batches through the planner's choice against the exact scan."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

rows, dim, k = 10_000_000, 768, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="L2")
ix.fill_synthetic_clustered(rows, seed=0, nclusters=4096, spread=0.1)
for Q in (1, 8, 32, 64, 200):
    qs = _native.Index(rt, dim, metric="L2")
    qs.fill_synthetic_clustered(Q, seed=0, nclusters=4096, spread=0.1, first_row=rows + 999)
    q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
    qs.close()
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out = {}
    for mode in ("auto", "exact"):
        ix.set_search_mode(mode)
        for _ in range(3):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        out[mode] = ((time.perf_counter() - t0) / 3, ix.last_search_stats(), orow.cpu().numpy().copy(), od.cpu().numpy().copy())
    same = np.array_equal(out["auto"][2], out["exact"][2]) and np.array_equal(out["auto"][3].view(np.uint32), out["exact"][3].view(np.uint32))
    print(f"Q {Q:4d}: auto {out['auto'][0] * 1e3:7.2f} ms {out['auto'][1]} | exact scan {out['exact'][0] * 1e3:7.2f} ms | same bits {same}", flush=True)
ix.close(); rt.close()
