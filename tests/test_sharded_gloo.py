"""CPU, world_size 2 and 3 over gloo: the row-range shard + all-gather + merge path is shard-count invariant."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from semcode_amd import _native
from semcode_amd.storage.sharded import shard_range

ROOT = Path(__file__).resolve().parent.parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_partition_the_rows():
    for n, w in [(10, 1), (10, 3), (80_000_000, 8), (5, 8), (0, 2)]:
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_range(80_000_000, 8, 3) == (30_000_000, 40_000_000)  # BASELINE config 4


def test_host_merge_tie_rule_and_missing_hits():
    # two shards, equal distances: lower global row id first; -1 rows are skipped
    d = np.array([[[1.0, 2.0, np.inf]], [[1.0, 1.5, 2.0]]], np.float32)
    r = np.array([[[7, 9, -1]], [[3, 20, 8]]], np.int64)
    md, mr = _native.topk_merge_host("L2", d, r)
    assert mr.tolist() == [[3, 7, 20]] and md.tolist() == [[1.0, 1.0, 1.5]]
    md, mr = _native.topk_merge_host("IP", -d, r)
    assert mr.tolist() == [[3, 7, 20]]
    md, mr = _native.topk_merge_host("L2", d[:1, :, 2:], r[:1, :, 2:])
    assert mr.tolist() == [[-1]] and np.isinf(md).all()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_matches_single_index(world, tmp_path):
    out = tmp_path / "result.txt"
    env = dict(os.environ, TEST_OUT=str(out), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(ROOT / "tests" / "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    assert out.read_text() == "OK"
