import sys
sys.path.insert(0, ".")
import torch
from semcode_amd import _native
rows, dim, nlist, nprobe, Q, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000), 3072, (int(sys.argv[2]) if len(sys.argv) > 2 else 1600), 64, 1024, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="L2", kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=nlist, spread=0.5)
qs = _native.Index(rt, dim, metric="L2")
qs.fill_synthetic_clustered(Q, seed=0, nclusters=nlist, spread=0.5, first_row=rows + 12345)
q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
qs.close()
ix.train(niter=4)
od = torch.empty((Q, k), dtype=torch.float32, device=dev)
orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
ix.set_search_mode("ivf_coarse")
for _ in range(3):
    ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
rt.synchronize()
print(ix.last_search_stats())
ix.close(); rt.close()
