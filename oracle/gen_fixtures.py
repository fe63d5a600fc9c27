"""Generate the committed golden vectors under tests/golden/.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_fixtures [knn] [synth] [md5] [encoder] [ivf]

The reference itself cannot be imported here (langchain / pymilvus absent) and
its tests hold no numerical vector for this path, so these fixtures are produced by independent
implementations available in the container:
  * knn_*      : numpy float64 brute force (ids, order) + the C restatement's f32 distances
  * synth_kat  : values of the integer-hash generator computed with Python big-int arithmetic
  * chunk_id   : hashlib.md5 known answers for IndexerService._make_chunk_id
                 (reference src/semcode/services/indexer.py:185-188)
  * encoder_*  : transformers' local BertModel class with seeded random weights (see gen_encoder())
  * ivf_*      : the deterministic IVF_FLAT build rule itself (a regression lock, see gen_ivf(): nothing independent exists)
"""
from __future__ import annotations

import hashlib
import json
import sys
from pathlib import Path

import numpy as np

from . import sc_oracle as orc

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"


def gen_knn() -> None:
    rng = np.random.default_rng(20251031)
    X = rng.standard_normal((4096, 64)).astype(np.float32)
    Q = rng.standard_normal((32, 64)).astype(np.float32)
    # plant exact ties: duplicate rows must come back in row order
    X[100] = X[7]
    X[3000] = X[7]
    Q[0] = X[7] + 0.01 * rng.standard_normal(64).astype(np.float32)
    out = {"X": X, "Q": Q}
    for metric in ("IP", "L2", "COSINE"):
        s64, r64 = orc.search_f64(X, Q, 10, metric)
        d32, r32 = orc.search(X, Q, 10, metric)
        # the fixture is only valid if both independent implementations agree on ids and order
        assert np.array_equal(r64, r32), f"{metric}: f64 brute force and C restatement disagree"
        np.testing.assert_allclose(d32, s64, rtol=2e-5, atol=2e-4)
        out[f"{metric}_rows"] = r64
        out[f"{metric}_dist"] = d32
        out[f"{metric}_score64"] = s64
    np.savez_compressed(GOLDEN / "knn_4096x64.npz", **out)
    print("wrote knn_4096x64.npz")


def gen_ivf() -> None:
    """IVF_FLAT build + probe of the deterministic build rule (oracle/ivf_oracle.py restates semcode_amd/csrc/sc_ivf.cpp) on the kNN
    fixture's rows.  There is nothing independent to generate these from (Milvus' k-means is random and absent): the file LOCKS
    the rule -- sampling, initialisation, Lloyd + re-seeding, assignment metric, probe order -- against silent change, and both the
    CPU restatement and the GPU build are checked against it."""
    from .ivf_oracle import IvfOracle

    d = np.load(GOLDEN / "knn_4096x64.npz")
    X, Q = d["X"], d["Q"]
    out = {}
    for metric in ("IP", "L2", "COSINE"):
        o = IvfOracle(X, metric, nlist=16, niter=6)
        dist, rows = o.search(Q, 10, 4)
        out[f"{metric}_centroids"] = o.centroids
        out[f"{metric}_assign"] = o.assign.astype(np.int32)
        out[f"{metric}_rows"] = rows
        out[f"{metric}_dist"] = dist
    np.savez_compressed(GOLDEN / "ivf_4096x64.npz", **out)
    print("wrote ivf_4096x64.npz")


M64 = (1 << 64) - 1


def _mix64(z: int) -> int:
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_py(seed: int, row: int, col: int, dim: int) -> float:
    """Big-int restatement of the generator (independent of the C code)."""
    key = _mix64((seed + 0x9E3779B97F4A7C15) & M64)
    ctr = ((row * dim + col) * 3) & M64
    total = 0
    for j in range(3):
        h = _mix64(key ^ ((((ctr + j) & M64) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & M64))
        total += (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)
    return float(np.float32(total - 393210) * np.float32(1.0 / 65536.0))


def gen_synth() -> None:
    cases = []
    for seed, row, col, dim in [(0, 0, 0, 768), (0, 1, 5, 768), (1, 9_999_999, 767, 768), (42, 123456789, 3071, 3072),
                                (7, 2**33 + 5, 63, 64), (2**63 + 11, 31, 0, 1)]:
        cases.append({"seed": seed, "row": row, "col": col, "dim": dim, "value": synth_py(seed, row, col, dim)})
    (GOLDEN / "synth_kat.json").write_text(json.dumps(cases, indent=1))
    print("wrote synth_kat.json")


def gen_md5() -> None:
    cases = []
    for repo, path, start, end in [("demo", "/tmp/workspace/demo/demo_src/example.py", 1, 2), ("r", "/w/r/a.py", 1, 200),
                                   ("r", "/w/r/a.py", 201, 400)]:
        text = f"{repo}:{path}:{start}:{end}"
        cases.append({"repo": repo, "path": path, "start": start, "end": end, "md5": hashlib.md5(text.encode("utf-8")).hexdigest()})
    # the three values quoted in SURVEY.md section 8a6 must come out of hashlib too
    assert cases[0]["md5"] == "132882290f50926ff2c7d796ed5dbc63"
    assert cases[1]["md5"] == "212daf327bfd98dd3dbd09c0e0933943"
    assert cases[2]["md5"] == "d74a91750bfc8f078573bc73e939d723"
    (GOLDEN / "chunk_id_kat.json").write_text(json.dumps(cases, indent=1))
    print("wrote chunk_id_kat.json")


def main(argv: list[str]) -> None:
    GOLDEN.mkdir(parents=True, exist_ok=True)
    what = set(argv) or {"knn", "synth", "md5", "encoder", "ivf"}
    if "knn" in what:
        gen_knn()
    if "ivf" in what:
        gen_ivf()
    if "synth" in what:
        gen_synth()
    if "md5" in what:
        gen_md5()
    if "encoder" in what:
        try:
            from .gen_encoder_fixtures import gen_encoder
        except ImportError:
            print("encoder fixtures: generator not present yet, skipped")
        else:
            gen_encoder()


if __name__ == "__main__":
    main(sys.argv[1:])
