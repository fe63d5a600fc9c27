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
