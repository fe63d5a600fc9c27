"""CPU: pin the oracle (oracle/sc_oracle.c) against the committed golden vectors and an independent float64 brute force."""
import hashlib
import json

import numpy as np
import pytest

from oracle import sc_oracle as orc
from oracle.gen_fixtures import synth_py


@pytest.fixture(scope="module")
def knn(golden):
    return np.load(golden / "knn_4096x64.npz")


@pytest.mark.parametrize("metric", ["IP", "L2", "COSINE"])
def test_search_matches_golden(knn, metric):
    dist, rows = orc.search(knn["X"], knn["Q"], 10, metric)
    assert np.array_equal(rows, knn[f"{metric}_rows"])  # integer ids and order: exact
    assert np.array_equal(dist.view(np.uint32), knn[f"{metric}_dist"].view(np.uint32))  # f32 distances: bit-exact
    np.testing.assert_allclose(dist, knn[f"{metric}_score64"], rtol=2e-5, atol=2e-4)  # vs float64 brute force


def test_tie_rule_lower_row_first(knn):
    # rows 7, 100, 3000 are identical; query 0 is a perturbation of row 7 => they lead, in row order
    for metric in ("IP", "L2", "COSINE"):
        _, rows = orc.search(knn["X"], knn["Q"][:1], 10, metric)
        got = [int(r) for r in rows[0] if r in (7, 100, 3000)]
        assert got == [7, 100, 3000], (metric, rows[0])
    _, rows = orc.search(knn["X"], knn["Q"][:1], 3, "L2")
    assert rows[0].tolist() == [7, 100, 3000]


@pytest.mark.parametrize("metric", ["IP", "L2", "COSINE"])
def test_search_vs_float64_random(metric):
    rng = np.random.default_rng(5)
    X = rng.standard_normal((3001, 100)).astype(np.float32)  # ragged: n % 16 != 0, dim % 64 != 0
    Q = rng.standard_normal((7, 100)).astype(np.float32)
    s64, r64 = orc.search_f64(X, Q, 13, metric)
    d, r = orc.search(X, Q, 13, metric)
    assert np.array_equal(r, r64)
    np.testing.assert_allclose(d, s64, rtol=2e-5, atol=2e-4)


def test_search_fewer_rows_than_k():
    rng = np.random.default_rng(6)
    X = rng.standard_normal((3, 64)).astype(np.float32)
    Q = rng.standard_normal((2, 64)).astype(np.float32)
    d, r = orc.search(X, Q, 5, "L2", row_base=1000)
    assert (r[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all()
    assert set(r[0, :3].tolist()) == {1000, 1001, 1002}
    d, r = orc.search(X, Q, 5, "IP")
    assert (d[:, 3:] == -np.inf).all()


def test_search_rows_subset():
    rng = np.random.default_rng(8)
    X = rng.standard_normal((500, 64)).astype(np.float32)
    q = rng.standard_normal(64).astype(np.float32)
    sub = np.arange(0, 500, 7)
    d, r = orc.search_rows(X, q, sub, 5, "L2")
    dfull, rfull = orc.search(X[sub], q[None, :], 5, "L2")
    assert np.array_equal(r, sub[rfull[0]])
    assert np.array_equal(d.view(np.uint32), dfull[0].view(np.uint32))


def test_sqnorm_and_dot_close_to_float64():
    rng = np.random.default_rng(9)
    for n in (64, 768, 3072):
        x = rng.standard_normal(n).astype(np.float32)
        q = rng.standard_normal(n).astype(np.float32)
        assert abs(orc.sqnorm(x) - float((x.astype(np.float64) ** 2).sum())) <= 1e-6 * n
        assert abs(orc.dot(x, q) - float(x.astype(np.float64) @ q.astype(np.float64))) <= 2e-6 * n
    # padding with zeros must not change either value
    x = rng.standard_normal(100).astype(np.float32)
    xp = orc.padded(x[None, :])[0]
    assert orc.sqnorm(xp) == orc.sqnorm(np.concatenate([xp, np.zeros(64, np.float32)]))


def test_synth_known_answers(golden):
    cases = json.loads((golden / "synth_kat.json").read_text())
    h = orc.lib()
    for c in cases:
        got = float(h.sc_oracle_synth(c["seed"], c["row"], c["col"], c["dim"]))
        assert got == c["value"], c
        assert synth_py(c["seed"], c["row"], c["col"], c["dim"]) == c["value"]


def test_synth_is_standard_normal_like_and_random_access():
    X = orc.synth(4000, 64, seed=3)
    assert abs(float(X.mean())) < 0.01 and abs(float(X.std()) - 1.0) < 0.01
    assert abs(float((X ** 3).mean())) < 0.03          # symmetric
    assert 2.7 < float((X ** 4).mean()) < 3.1          # Irwin-Hall(12) kurtosis = 2.9
    # any row can be regenerated on its own, and shards concatenate (first_row offset)
    assert np.array_equal(orc.synth_rows([17, 3999], 64, 3), X[[17, 3999]])
    assert np.array_equal(orc.synth(100, 64, 3, first_row=1000), X[1000:1100])
    # padding columns are zero and do not shift the values
    Xp = orc.synth(10, 50, seed=4, ld=64)
    assert (Xp[:, 50:] == 0).all() and np.array_equal(Xp[:, :50], orc.synth(10, 50, seed=4))


def test_chunk_id_known_answers(golden):
    # reference: IndexerService._make_chunk_id, src/semcode/services/indexer.py:185-188
    for c in json.loads((golden / "chunk_id_kat.json").read_text()):
        text = f"{c['repo']}:{c['path']}:{c['start']}:{c['end']}"
        assert hashlib.md5(text.encode("utf-8")).hexdigest() == c["md5"]
