"""top_k beyond 64 for a batch: the int8 stage (512 candidates: k <= 256) against the exact scan (one pass per 16 queries)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

rows, dim = 10_000_000, 768
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="L2")
ix.fill_synthetic(rows, seed=0)
for Q, k in ((256, 100), (64, 100), (1024, 65), (1024, 128), (8, 100), (1, 80), (256, 200), (1024, 10)):
    q = torch.empty((Q, dim), dtype=torch.float32, device=dev)
    rt.synth_fill_dev(q.data_ptr(), Q, dim, dim, seed=7)
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out = {}
    for mode in ("auto", "exact"):
        ix.set_search_mode(mode)
        ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        t0 = time.perf_counter()
        n = 3 if mode == "auto" else 1
        for _ in range(n):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        out[mode] = ((time.perf_counter() - t0) / n, ix.last_search_stats(), orow.cpu().numpy().copy(), od.cpu().numpy().copy())
    same = np.array_equal(out["auto"][2], out["exact"][2]) and np.array_equal(out["auto"][3].view(np.uint32), out["exact"][3].view(np.uint32))
    print(f"Q {Q:5d} k {k:4d}: auto {out['auto'][0] * 1e3:8.2f} ms {out['auto'][1]} | exact scan {out['exact'][0] * 1e3:8.2f} ms | same bits {same}", flush=True)
ix.close(); rt.close()
