"""ms per batch of the exhaustive search over 10M x 768 for a range of batch sizes (one process, one index)."""
import sys
import time

sys.path.insert(0, ".")
import torch

from semcode_amd import _native

import os
rows, dim, k = int(os.environ.get('SC_Q_ROWS', 10_000_000)), 768, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="L2")
ix.fill_synthetic(rows, seed=0)
import os
if os.environ.get("SC_Q_MODE"):
    ix.set_search_mode(os.environ["SC_Q_MODE"])  # e.g. "batched": the coarse stages for every batch size
q = torch.empty((1024, dim), dtype=torch.float32, device=dev)
rt.synth_fill_dev(q.data_ptr(), 1024, dim, dim, seed=1)
od = torch.empty((1024, k), dtype=torch.float32, device=dev)
orow = torch.empty((1024, k), dtype=torch.int64, device=dev)
for nq in [int(a) for a in sys.argv[1:]] or [17, 32, 64, 128, 129, 256, 512, 1024]:
    for _ in range(3):
        ix.search_dev(q.data_ptr(), nq, k, od.data_ptr(), orow.data_ptr())
    rt.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search_dev(q.data_ptr(), nq, k, od.data_ptr(), orow.data_ptr())
    rt.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = ix.last_search_stats()
    print(f"Q={nq:5d}  {dt * 1e3:7.2f} ms   {nq / dt:9.0f} QPS   path {st['path']} coarse {st.get('coarse_bits')} uncertified {st['uncertified']}", flush=True)
ix.close()
rt.close()
