"""The reference's own index parameters (IVF_FLAT, IP, nlist 128, nprobe 16; milvus_store.py:76-84,144) at 768 dimensions: what the
auto planner picks for a batch and what each path costs on the same index -- exhaustive (exact), list-major exact probe, list-major
behind the int8 coarse stage."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from semcode_amd import _native

dim, nlist, nprobe, k = 768, 128, 16, 5
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
for rows in (1_000_000, 10_000_000):
    ix = _native.Index(rt, dim, metric="IP", kind="IVF_FLAT", nlist=nlist)
    ix.fill_synthetic_clustered(rows, seed=0, nclusters=1024, spread=0.5)
    t0 = time.perf_counter()
    ix.train(niter=8)
    rt.synchronize()
    sizes = ix.ivf_info()["list_sizes"]
    print(f"{rows} x {dim}, IP, nlist {nlist}: train {time.perf_counter() - t0:.1f} s, list sizes {int(sizes.min())} / {int(np.median(sizes))} / {int(sizes.max())}", flush=True)
    for Q in (64, 256, 1024):
        qs = _native.Index(rt, dim, metric="IP")
        qs.fill_synthetic_clustered(Q, seed=0, nclusters=1024, spread=0.5, first_row=rows + 777)
        q = torch.from_numpy(qs.get_rows(0, Q)).to(dev)
        qs.close()
        od = torch.empty((Q, k), dtype=torch.float32, device=dev)
        orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
        out = {}
        for mode in ("auto", "ivf_listmajor", "ivf_coarse", "batched"):
            ix.set_search_mode(mode)
            npb = nlist if mode == "batched" else nprobe
            for _ in range(2):
                ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=npb)
            rt.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=npb)
            rt.synchronize()
            dt = (time.perf_counter() - t0) / 4
            st = ix.last_search_stats()
            out[mode] = (dt, st["path"], st["uncertified"], orow.cpu().numpy().copy(), od.cpu().numpy().copy())
        same = np.array_equal(out["ivf_listmajor"][3], out["ivf_coarse"][3]) and np.array_equal(out["ivf_listmajor"][4].view(np.uint32), out["ivf_coarse"][4].view(np.uint32))
        recall = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(out["ivf_coarse"][3].tolist(), out["batched"][3].tolist())]))
        print(f"  Q {Q:5d}: auto -> {out['auto'][1]:14s} {out['auto'][0] * 1e3:7.2f} ms | exact list-major {out['ivf_listmajor'][0] * 1e3:7.2f} ms | coarse {out['ivf_coarse'][0] * 1e3:7.2f} ms "
              f"(to exact probe {out['ivf_coarse'][2]}, same bits {same}) | exhaustive {out['batched'][1]} {out['batched'][0] * 1e3:7.2f} ms | recall@{k} of probing {recall:.3f}", flush=True)
    ix.close()
rt.close()
