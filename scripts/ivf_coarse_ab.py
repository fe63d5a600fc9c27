"""Config 5 (10M x 3072, nlist 4096, nprobe 64, batch 1024, L2 top-10): exact list-major probing vs the int8 coarse stage in front
of it, one index, one process.  Results must be the same bits."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from semcode_amd import _native

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
metric = sys.argv[2] if len(sys.argv) > 2 else "L2"
dim, nlist, nprobe, Q, k = 3072, 4096, 64, 1024, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric=metric, kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=nlist, spread=0.5)
qs = _native.Index(rt, dim, metric=metric)
qs.fill_synthetic_clustered(Q, seed=0, nclusters=nlist, spread=0.5, first_row=rows + 12345)
q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
qs.close()
t0 = time.perf_counter()
ix.train(niter=10)
rt.synchronize()
print(f"{metric}: train {time.perf_counter() - t0:.1f} s", flush=True)
od = torch.empty((Q, k), dtype=torch.float32, device=dev)
orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
res = {}
for mode in ("ivf_listmajor", "ivf_coarse", "ivf_listmajor", "ivf_coarse"):
    ix.set_search_mode(mode)
    for _ in range(2):
        ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    rt.set_profiling(True)
    rt.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    dt = (time.perf_counter() - t0) / 5
    k_ms, k_n = rt.profile_read(0)
    m_ms, m_n = rt.profile_read(1)
    rt.set_profiling(False)
    st, pst = ix.last_search_stats(), ix.last_probe_stats()
    res[mode] = (od.cpu().numpy().copy(), orow.cpu().numpy().copy())
    print(f"{mode:14s} {dt * 1e3:7.2f} ms / batch  {Q / dt:9.0f} QPS   scan kernels {k_ms / 5:6.2f} ms  select {m_ms / 5:5.2f} ms   uncertified {st['uncertified']:4d}   "
          f"streamed {pst['streamed_rows'] * 3072 / 1e9:6.1f} G row-bytes(int8) / {pst['streamed_rows'] * 3072 * 4 / 1e9:6.1f} GB f32, unique {pst['unique_rows'] * 3072 * 4 / 1e9:6.1f} GB f32",
          flush=True)
a, b = res["ivf_listmajor"], res["ivf_coarse"]
print("same ids:", np.array_equal(a[1], b[1]), " same distance bits:", np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), flush=True)
ix.close()
rt.close()
