"""Rehearsal of the N > 1 bench path on a 1-GPU box: 2 ranks share device 0 over gloo; rank 0 checks that the merged
result of the two 1M-row shards equals a single 2M-row index."""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, ".")
from semcode_amd import _native
from semcode_amd.storage.sharded import ShardedSearcher, shard_range

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rt = _native.Runtime(0)
N, D = 2_000_000, 768
s, e = shard_range(N, world, rank)
ix = _native.Index(rt, D, metric="L2", row_base=s)
ix.fill_synthetic(e - s, seed=0, first_row=s)
q = torch.empty((64, D), dtype=torch.float32, device="cuda")
rt.synth_fill_dev(q.data_ptr(), 64, D, D, seed=1)
rt.synchronize()
Q = q.cpu().numpy()
d, r = ShardedSearcher(ix, "L2", device="cpu").search(Q, k=10)
if rank == 0:
    full = _native.Index(rt, D, metric="L2")
    full.fill_synthetic(N, seed=0)
    fd, fr = full.search(Q, k=10)
    ok = np.array_equal(r, fr) and np.array_equal(d.view(np.uint32), fd.view(np.uint32))
    print("sharded == single:", ok, flush=True)
    full.close()
    assert ok
dist.barrier()
ix.close(); rt.close()
dist.destroy_process_group()
