"""Row-range sharding of the corpus over the GPUs of one node (one process per GPU).

Not in the reference (single Milvus server, no parallelism: SURVEY.md section 2); this is the
north_star's multi-GPU scheme: rank r holds rows [r*ceil(N/R), ...), every rank searches the same
replicated queries on its shard, the per-shard [Q, k] results (f32 distance + i64 GLOBAL row id) are
exchanged with ONE all-gather and merged with the same (distance, lower row id) rule, so results do
not depend on the shard count.

The exchange runs behind the C ABI: `sc_index_search_sharded` = shard search + RCCL all-gather over xGMI on
the runtime's stream + host merge, one call per rank (`_native.Comm`, include/semcode_hip.h "communicator").
No framework sits on that path; `make_comm` only needs some way to hand rank 0's 128-byte rendezvous id to
the other ranks (torch.distributed's store when a process group exists, or the caller's own channel).
A torch.distributed process group (gloo) remains supported as the exchange for the CPU tests and for
rehearsing several ranks on one device, where RCCL cannot run (one rank per GPU).
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


def make_comm(runtime: Any, rank: Optional[int] = None, world: Optional[int] = None, unique_id: Optional[bytes] = None, group: Any = None) -> Any:
    """The native RCCL communicator of this rank (`_native.Comm`).  The rendezvous id is created on rank 0 and, unless the caller
    passes it (`unique_id`, e.g. read from a file or an environment variable its launcher set), travels through the
    torch.distributed process group that the launcher initialised (any backend: it is a 128-byte control message)."""
    from .. import _native

    if unique_id is None:
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("make_comm: pass unique_id, or initialise a torch.distributed process group to carry it")
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        box = [_native.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        unique_id = box[0]
    if rank is None or world is None:
        raise ValueError("make_comm: rank and world are required with an explicit unique_id")
    return _native.Comm(runtime, rank, world, unique_id)


class ShardedSearcher:
    """Search a row-sharded collection.  `index` is this rank's shard (created with row_base = shard start).

    comm: a `_native.Comm` -> the exchange is RCCL behind the C ABI (the product path on GPUs);
    otherwise `group` (or the default torch.distributed group) carries it on the host (gloo: CPU tests, rehearsal)."""

    def __init__(self, index: Any, metric: str, group: Any = None, device: Optional[Any] = None, comm: Any = None) -> None:
        self.index = index
        self.metric = metric
        self.group = group
        self.device = device
        self.comm = comm

    # ------------------------------------------------------------------ IVF_FLAT build
    def train(self, niter: int = 10, src: int = 0) -> None:
        """IVF_FLAT over a sharded collection: rank `src` runs k-means on ITS shard, its centroids are
        broadcast (the one collective of the build; 50 MB at nlist 4096 x 3072), every rank assigns its own rows to them.  Probing
        then looks at the same lists on every shard, so the merged result is that of one index with these centroids."""
        if self.comm is not None:
            self.index.train_sharded(self.comm, niter=niter, root=src)
            return
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

    # ------------------------------------------------------------------ search
    def search(self, queries: np.ndarray, k: int = 10, nprobe: int = 16) -> Tuple[np.ndarray, np.ndarray]:
        """queries [Q, dim] (identical on every rank) -> (dist [Q, k], rows [Q, k] global ids), on every rank."""
        if self.comm is not None:
            return self.index.search_sharded(self.comm, queries, k=k, nprobe=nprobe)
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

    def search_dev(self, q_ptr: int, Q: int, k: int, all_dist_ptr: int, all_rows_ptr: int, nprobe: int = 16) -> None:
        """Device-resident variant (native communicator only): queries at q_ptr, every shard's [Q, k] result into the caller's
        device arrays [world, Q, k]; nothing leaves the device until the caller copies the gathered arrays out for the merge."""
        if self.comm is None:
            raise RuntimeError("search_dev needs the native communicator (make_comm)")
        self.index.search_sharded_dev(self.comm, q_ptr, Q, k, all_dist_ptr, all_rows_ptr, nprobe=nprobe)
