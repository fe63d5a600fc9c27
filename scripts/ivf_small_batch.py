"""Config-5 index (10M x 3072, nlist 4096, nprobe 64, L2): small batches through the per-query probe, the exact list-major probe and
the int8 coarse stage -- where does the coarse stage start to pay?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

# usage: ivf_small_batch.py [rows dim nlist nprobe metric]   (default: config 5; "10000000 768 128 16 IP" = the reference's parameters)
a = sys.argv[1:]
rows, dim, nlist, nprobe, k = (int(a[0]), int(a[1]), int(a[2]), int(a[3]), 10) if len(a) >= 4 else (10_000_000, 3072, 4096, 64, 10)
metric = a[4] if len(a) >= 5 else "L2"
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric=metric, kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=max(nlist, 1024), spread=0.5)
ix.train(niter=10)
for Q in (1, 2, 4, 8, 16, 32, 48):
    qs = _native.Index(rt, dim, metric=metric)
    qs.fill_synthetic_clustered(Q, seed=0, nclusters=max(nlist, 1024), spread=0.5, first_row=rows + 4242)
    q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
    qs.close()
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out = {}
    for mode in ("auto", "ivf", "ivf_listmajor", "ivf_coarse"):
        ix.set_search_mode(mode)
        for _ in range(2):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
        rt.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
        rt.synchronize()
        out[mode] = ((time.perf_counter() - t0) / 5, ix.last_search_stats()["path"], orow.cpu().numpy().copy(), od.cpu().numpy().copy())
    same = all(np.array_equal(out["ivf"][2], out[m][2]) and np.array_equal(out["ivf"][3].view(np.uint32), out[m][3].view(np.uint32)) for m in ("ivf_listmajor", "ivf_coarse", "auto"))
    print(f"{rows} x {dim} nlist {nlist} nprobe {nprobe} {metric}  Q {Q:3d}: auto -> {out['auto'][1]:14s} {out['auto'][0] * 1e3:6.2f} ms | per query {out['ivf'][0] * 1e3:6.2f} | list-major {out['ivf_listmajor'][0] * 1e3:6.2f} ({out['ivf_listmajor'][1]}) | coarse {out['ivf_coarse'][0] * 1e3:6.2f} ({out['ivf_coarse'][1]}) | same bits {same}", flush=True)
ix.close(); rt.close()
