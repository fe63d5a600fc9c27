"""Is the int8 coarse main loop waiting for HBM?  The same kernel over a corpus that fits the 256 MB MALL (many queries, so that the
launch still has thousands of tiles) against the 10M-row corpus: per-tile stamps via SC_COARSE_TRACE=5 SC_COARSE_TRACE_MIN=4000."""
import sys
sys.path.insert(0, ".")
import torch
from semcode_amd import _native

rows, Q = int(sys.argv[1]), int(sys.argv[2])
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, 768, metric="L2")
ix.fill_synthetic(rows, seed=0)
qs = _native.Index(rt, 768, metric="L2")
qs.fill_synthetic(Q, seed=1)
q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
qs.close()
ix.set_search_mode("batched")
ix.set_coarse_stage(8)
od = torch.empty((Q, 10), dtype=torch.float32, device=dev)
orow = torch.empty((Q, 10), dtype=torch.int64, device=dev)
for _ in range(3):
    ix.search_dev(q.data_ptr(), Q, 10, od.data_ptr(), orow.data_ptr())
rt.synchronize()
print(rows, Q, ix.last_search_stats(), flush=True)
ix.close(); rt.close()
