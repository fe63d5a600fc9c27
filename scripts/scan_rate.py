"""Streaming rate of the exact scan's variants on a flat index (tuning aid): resident queries vs streamed queries.
    python3 scripts/scan_rate.py [dim rows]"""
import os
import sys
import time

sys.path.insert(0, ".")
from semcode_amd import _native

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rt = _native.Runtime(0)
ix = _native.Index(rt, dim, metric="L2")
ix.fill_synthetic(rows, seed=1)
qs = _native.Index(rt, dim, metric="L2")
qs.fill_synthetic(16, seed=2)
Q = qs.get_rows(0, 16)
ix.set_search_mode("exact")
gb = rows * ((dim + 63) // 64 * 64) * 4 / 1e9


def timed(nq, env):
    if env is None:
        os.environ.pop("SC_SCAN_QSTREAM", None)
    else:
        os.environ["SC_SCAN_QSTREAM"] = env
    ix.search(Q[:nq], k=10)
    t = time.time()
    for _ in range(5):
        ix.search(Q[:nq], k=10)
    ms = (time.time() - t) / 5 * 1e3
    print(f"dim={dim} rows={rows} Q={nq:2d} SC_SCAN_QSTREAM={env}: {ms:8.3f} ms per search = {gb / ms:6.2f} TB/s per pass-equivalent of {gb:.1f} GB", flush=True)


for nq, env in ((1, "0"), (6, "0"), (16, "0"), (16, "1"), (6, "1"), (16, None)):
    timed(nq, env)
