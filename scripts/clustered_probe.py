"""Batched exhaustive search on a CLUSTERED 10M x 768 corpus (sc_index_fill_synthetic_clustered): ms per 1024-query batch,
queries handed from the int8 to the bf16 stage, queries left to the exact scan -- next to the Gaussian corpus of the headline."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from semcode_amd import _native

rows, dim, Q, k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768, 1024, 10
stream = torch.cuda.Stream()
rt = _native.Runtime(device=0, stream=stream.cuda_stream)
dev = torch.device("cuda", 0)
for name, ncl, spread in (("gaussian", 0, 0.0), ("clustered 4096 x 0.5", 4096, 0.5), ("clustered 4096 x 0.1", 4096, 0.1), ("clustered 256 x 0.5", 256, 0.5)):
    ix = _native.Index(rt, dim, metric="L2")
    qsrc = _native.Index(rt, dim, metric="L2")
    if ncl:
        ix.fill_synthetic_clustered(rows, seed=0, nclusters=ncl, spread=spread)
        qsrc.fill_synthetic_clustered(Q, seed=0, nclusters=ncl, spread=spread, first_row=rows + 12345)
    else:
        ix.fill_synthetic(rows, seed=0)
        qsrc.fill_synthetic(Q, seed=1)
    q = torch.from_numpy(qsrc.get_rows(0, Q)).to(dev)
    qsrc.close()
    od = torch.empty((Q, k), dtype=torch.float32, device=dev)
    orow = torch.empty((Q, k), dtype=torch.int64, device=dev)
    for stage in (0, 8, 16):
        ix.set_coarse_stage(stage)
        for _ in range(2):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.search_dev(q.data_ptr(), Q, k, od.data_ptr(), orow.data_ptr())
        rt.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = ix.last_search_stats()
        print(f"{name:22s} stage setting {stage:2d}, first stage {'int8' if st.get('coarse_bits') == 8 else 'bf16'}{' (wide)' if st.get('wide') else '       '}: {dt * 1e3:8.2f} ms / batch   collect pass {st.get('collect_resolved', 0):4d} of "
              f"{st.get('collect_tried', 0):4d}   handed to bf16 {st.get('handed_to_bf16', 0):4d}   uncertified (exact scan) {st['uncertified']:4d}", flush=True)
    ix.close()
rt.close()
