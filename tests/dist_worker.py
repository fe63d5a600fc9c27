"""Worker for tests/test_sharded_gloo.py (run under torch.distributed.run, gloo, CPU).

Each rank holds one row-range shard of a seeded corpus behind an index object with the
_native.Index search surface (here: the CPU oracle, since there is no GPU), runs the product's
ShardedSearcher, and rank 0 checks the merged result against the oracle over the whole corpus."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch.distributed as dist  # noqa: E402

from oracle import sc_oracle as orc  # noqa: E402
from semcode_amd.storage.sharded import ShardedSearcher, shard_range  # noqa: E402


class OracleShard:
    def __init__(self, X, metric, row_base):
        self.X, self.metric, self.row_base = X, metric, row_base

    def search(self, q, k=10, nprobe=16):
        return orc.search(self.X, q, k, self.metric, row_base=self.row_base)


class OracleIvfShard:
    """A shard with the IVF surface ShardedSearcher.train() drives (train / ivf_info / assign_lists), on the CPU restatement."""

    def __init__(self, X, metric, row_base, nlist):
        self.X, self.metric, self.row_base, self.nlist = X, metric, row_base, nlist
        self.centroids = self.lists = None

    def train(self, niter=10):
        from oracle.ivf_oracle import IvfOracle

        o = IvfOracle(self.X, self.metric, self.nlist, niter)
        self.centroids, self.lists = o.centroids, o.lists

    def ivf_info(self):
        return {"nlist": len(self.centroids), "centroids": self.centroids}

    def assign_lists(self, centroids):
        from oracle.ivf_oracle import _assign_metric, _nearest

        self.centroids = np.ascontiguousarray(centroids, np.float32)
        a = _nearest(self.centroids, self.X, _assign_metric(self.metric)) if len(self.X) else np.zeros(0, np.int64)
        self.lists = [np.nonzero(a == c)[0] for c in range(len(self.centroids))]

    def search(self, q, k=10, nprobe=16):
        _, probe = orc.search(self.centroids, q, nprobe, self.metric)
        dist_, rows = np.empty((len(q), k), np.float32), np.empty((len(q), k), np.int64)
        for i in range(len(q)):
            cand = np.sort(np.concatenate([self.lists[c] for c in probe[i] if c >= 0]))
            dist_[i], rows[i] = orc.search_rows(self.X, q[i], cand, k, self.metric)
        rows[rows >= 0] += self.row_base
        return dist_, rows


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, D, K = int(os.environ.get("TEST_ROWS", "5003")), 96, 10
    Q = orc.synth(9, D, seed=2)
    ok = True
    for metric in ("L2", "IP", "COSINE"):
        s, e = shard_range(N, world, rank)
        shard = OracleShard(orc.synth(e - s, D, seed=1, first_row=s), metric, s)
        d, r = ShardedSearcher(shard, metric).search(Q, k=K)
        if rank == 0:
            fd, fr = orc.search(orc.synth(N, D, seed=1), Q, K, metric)
            ok &= bool(np.array_equal(r, fr) and np.array_equal(d.view(np.uint32), fd.view(np.uint32)))
    # IVF_FLAT over shards: rank 0 trains, its centroids are broadcast, every rank assigns its rows; the merged probe result must
    # equal the probe result of ONE index over all rows with the same centroids
    s, e = shard_range(4000, world, rank)
    Xs = orc.synth_clustered(4000, D, 5, 12, 0.4)
    ivf = OracleIvfShard(Xs[s:e], "L2", s, nlist=8)
    searcher = ShardedSearcher(ivf, "L2")
    searcher.train(niter=4)
    d, r = searcher.search(Q, k=K, nprobe=3)
    if rank == 0:
        whole = OracleIvfShard(Xs, "L2", 0, nlist=8)
        whole.assign_lists(ivf.centroids)
        fd, fr = whole.search(Q, K, 3)
        ok &= bool(np.array_equal(r, fr) and np.array_equal(d.view(np.uint32), fd.view(np.uint32)))
    # a shard may be empty or shorter than k
    s, e = shard_range(world + 1, world, rank)
    tiny = OracleShard(orc.synth(e - s, D, seed=3, first_row=s), "L2", s)
    d, r = ShardedSearcher(tiny, "L2").search(Q, k=K)
    if rank == 0:
        fd, fr = orc.search(orc.synth(world + 1, D, seed=3), Q, K, "L2")
        ok &= bool(np.array_equal(r, fr) and np.array_equal(d.view(np.uint32), fd.view(np.uint32)))
        Path(os.environ["TEST_OUT"]).write_text("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
