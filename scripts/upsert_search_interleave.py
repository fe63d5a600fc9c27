"""What a search costs right after rows were appended to a trained IVF_FLAT index (10M x 768, the reference's parameters): the lists are
re-laid out before the search (sc_ivf_refresh_locked)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from semcode_amd import _native

rows, dim, nlist, nprobe, k, Q = 10_000_000, 768, 128, 16, 5, 32
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
ix = _native.Index(rt, dim, metric="IP", kind="IVF_FLAT", nlist=nlist)
ix.fill_synthetic_clustered(rows, seed=0, nclusters=1024, spread=0.5)
ix.train(niter=6)
qs = _native.Index(rt, dim, metric="IP")
qs.fill_synthetic_clustered(Q + 256, seed=0, nclusters=1024, spread=0.5, first_row=rows + 777)
allq = qs.get_rows(0, Q + 256)
qs.close()
q = torch.from_numpy(allq[:Q]).to(dev)
newrows = allq[Q:]
od = torch.empty((Q, k), dtype=torch.float32, device=dev)
orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
def search():
    rt.synchronize()
    t0 = time.perf_counter()
    ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr(), nprobe=nprobe)
    rt.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(3):
    search()
print(f"steady search of {Q} queries: {search():.2f} ms ({ix.last_search_stats()['path']})", flush=True)
for it in range(4):
    t0 = time.perf_counter()
    ix.add(newrows)
    rt.synchronize()
    t_add = (time.perf_counter() - t0) * 1e3
    t1 = search()
    t2 = search()
    print(f"append 256 rows {t_add:.2f} ms -> first search {t1:.2f} ms ({ix.last_search_stats()['path']}), second {t2:.2f} ms", flush=True)
# rows of the lists overwritten: with their own content (a re-index of unchanged files: same chunk ids, same vectors), then with new content
rng = np.random.default_rng(0)
for what in ("same content", "same content", "new content", "new content"):
    tgt = np.sort(rng.choice(rows, size=128, replace=False)).astype(np.int64)
    vec = np.stack([ix.get_rows(int(r), 1)[0] for r in tgt]) if what == "same content" else newrows[:128]
    t0 = time.perf_counter()
    ix.overwrite(vec, tgt)
    rt.synchronize()
    t_ow = (time.perf_counter() - t0) * 1e3
    t1 = search()
    t2 = search()
    print(f"overwrite 128 rows ({what}) {t_ow:.2f} ms -> first search {t1:.2f} ms ({ix.last_search_stats()}), second {t2:.2f} ms", flush=True)
ix.close()
# the same on a FLAT index (the batched path's int8 / bf16 shadows follow the rows)
fx = _native.Index(rt, dim, metric="IP")
fx.fill_synthetic_clustered(rows, seed=0, nclusters=1024, spread=0.5)
def fsearch():
    rt.synchronize()
    t0 = time.perf_counter()
    fx.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
    rt.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(3):
    fsearch()
print(f"FLAT: steady search of {Q} queries: {fsearch():.2f} ms ({fx.last_search_stats()['path']})", flush=True)
for what in ("overwrite", "overwrite", "append", "append"):
    t0 = time.perf_counter()
    if what == "overwrite":
        fx.overwrite(newrows[:128], np.sort(rng.choice(rows, size=128, replace=False)).astype(np.int64))
    else:
        fx.add(newrows[:128])
    rt.synchronize()
    t_ow = (time.perf_counter() - t0) * 1e3
    t1 = fsearch()
    t2 = fsearch()
    print(f"FLAT: {what} 128 rows {t_ow:.2f} ms -> first search {t1:.2f} ms ({fx.last_search_stats()['path']}), second {t2:.2f} ms", flush=True)
fx.close(); rt.close()
