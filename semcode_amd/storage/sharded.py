"""Row-range sharding of the corpus over the GPUs of one node (one process per GPU).

Not in the reference (single Milvus server, no parallelism: SURVEY.md section 2); this is the
north_star's multi-GPU scheme: rank r holds rows [r*ceil(N/R), ...), every rank searches the same
replicated queries on its shard, the per-shard [Q, k] results (f32 distance + i64 GLOBAL row id) are
exchanged with ONE all-gather (RCCL over xGMI when the process group is "nccl"; gloo in the CPU tests)
and merged with the same (distance, lower row id) rule, so results do not depend on the shard count.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [start, end) owned by `rank`: contiguous blocks of ceil(n_total / world) rows."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world / rank")
    per = -(-n_total // world)
    start = min(rank * per, n_total)
    return start, min(start + per, n_total)


class ShardedSearcher:
    """Search a row-sharded collection.  `index` is this rank's shard (created with row_base = shard start)."""

    def __init__(self, index: Any, metric: str, group: Any = None, device: Optional[Any] = None) -> None:
        self.index = index
        self.metric = metric
        self.group = group
        self.device = device

    def train(self, niter: int = 10, src: int = 0) -> None:
        """IVF_FLAT over a sharded collection: rank `src` runs k-means on ITS shard, its centroids are
        broadcast (the one collective of the build; 50 MB at nlist 4096 x 3072), every rank assigns its own rows to them.  Probing
        then looks at the same lists on every shard, so the merged result is that of one index with these centroids."""
        import torch
        import torch.distributed as dist

        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            self.index.train(niter=niter)
            return
        rank = dist.get_rank(self.group)
        dev = self.device or ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        if rank == src:
            self.index.train(niter=niter)
            cent = np.ascontiguousarray(self.index.ivf_info()["centroids"], dtype=np.float32)
            shape = torch.tensor(list(cent.shape), dtype=torch.int64, device=dev)
        else:
            shape = torch.zeros(2, dtype=torch.int64, device=dev)
        dist.broadcast(shape, src=src, group=self.group)
        nlist, dim = (int(v) for v in shape.tolist())
        t = torch.from_numpy(cent).to(dev) if rank == src else torch.empty((nlist, dim), dtype=torch.float32, device=dev)
        dist.broadcast(t, src=src, group=self.group)
        if rank != src:
            self.index.assign_lists(t.cpu().numpy())

    def search(self, queries: np.ndarray, k: int = 10, nprobe: int = 16) -> Tuple[np.ndarray, np.ndarray]:
        """queries [Q, dim] (identical on every rank) -> (dist [Q, k], rows [Q, k] global ids), on every rank."""
        import torch
        import torch.distributed as dist

        from .. import _native

        d, r = self.index.search(queries, k=k, nprobe=nprobe)
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return d, r
        world = dist.get_world_size(self.group)
        dev = self.device or ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        td = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
        tr = torch.from_numpy(np.ascontiguousarray(r)).to(dev)
        all_d = [torch.empty_like(td) for _ in range(world)]
        all_r = [torch.empty_like(tr) for _ in range(world)]
        dist.all_gather(all_d, td, group=self.group)  # the path's only exchange step
        dist.all_gather(all_r, tr, group=self.group)
        return _native.topk_merge_host(self.metric, torch.stack(all_d).cpu().numpy(), torch.stack(all_r).cpu().numpy())
