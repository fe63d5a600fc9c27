"""GPU: parity of the HIP distance scan + top-k (through the C ABI) against the CPU oracle.

Bar: row ids and order exact; f32 distances bit-exact against the oracle's canonical summation order
(oracle/sc_oracle.c), and within 2e-5 rel of a float64 brute force.
"""
import numpy as np
import pytest

from oracle import sc_oracle as orc
from semcode_amd import _native

pytestmark = pytest.mark.gpu

METRICS = ["IP", "L2", "COSINE"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check_exact(ix, X, Q, k, metric, row_base=0):
    d, r = ix.search(Q, k=k)
    od, orow = orc.search(X, Q, k, metric, row_base=row_base)
    assert np.array_equal(r, orow), f"{metric}: row ids / order differ"
    assert np.array_equal(bits(d), bits(od)), f"{metric}: distances not bit-exact (max abs diff {np.abs(d - od).max()})"
    return d, r


@pytest.mark.parametrize("metric", METRICS)
def test_golden_knn(rt, golden, metric):
    g = np.load(golden / "knn_4096x64.npz")
    ix = _native.Index(rt, 64, metric=metric)
    ix.add(g["X"])
    d, r = ix.search(g["Q"], k=10)
    assert np.array_equal(r, g[f"{metric}_rows"])
    assert np.array_equal(bits(d), bits(g[f"{metric}_dist"]))
    np.testing.assert_allclose(d, g[f"{metric}_score64"], rtol=2e-5, atol=2e-4)
    ix.close()


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("nq", [1, 5, 16, 33])
def test_seeded_100k_x_768(rt, metric, nq):
    X = orc.synth(100_000, 768, seed=11)
    Q = orc.synth(nq, 768, seed=12)
    ix = _native.Index(rt, 768, metric=metric)
    ix.add(X)
    check_exact(ix, X, Q, 10, metric)
    ix.close()


@pytest.mark.parametrize("n", [0, 1, 3, 15, 16, 17, 63, 64, 65, 1000, 16385])
def test_ragged_row_counts(rt, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 100)).astype(np.float32)  # dim % 64 != 0 -> padded stride
    Q = rng.standard_normal((3, 100)).astype(np.float32)
    ix = _native.Index(rt, 100, metric="L2", row_base=500)
    if n:
        ix.add(X)
    d, r = ix.search(Q, k=5)
    if n == 0:
        assert (r == -1).all() and np.isinf(d).all()
    else:
        od, orow = orc.search(X, Q, 5, "L2", row_base=500)
        assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
    ix.close()


@pytest.mark.parametrize("k", [1, 2, 10, 48, 49, 100, 257, 1024])
def test_k_range(rt, k):
    X = orc.synth(20_000, 128, seed=21)
    Q = orc.synth(4, 128, seed=22)
    ix = _native.Index(rt, 128, metric="IP")
    ix.add(X)
    check_exact(ix, X, Q, k, "IP")
    ix.close()


@pytest.mark.parametrize("metric", METRICS)
def test_streamed_query_variant_is_bit_identical(rt, metric, monkeypatch):
    """Long rows: the 16 queries of a pass are streamed through a shared LDS ring instead of living in LDS
    (scan_exact.hip, QS).  Forced on at dimensions where the resident variant is the default, it must return the same bits:
    ragged row counts (waves with unequal tile counts, workgroups without tiles), 1..16 queries, a second group."""
    for n, dim, nq, k in ((20_001, 768, 16, 10), (20_001, 768, 5, 10), (1, 768, 3, 1), (17, 100, 16, 5), (63, 100, 1, 5),
                          (4097, 100, 33, 10), (70_000, 256, 7, 48)):
        X = orc.synth(n, dim, seed=31 + n)
        Q = orc.synth(nq, dim, seed=32 + nq)
        ix = _native.Index(rt, dim, metric=metric)
        ix.add(X)
        ix.set_search_mode("exact")
        monkeypatch.setenv("SC_SCAN_QSTREAM", "0")
        d0, r0 = ix.search(Q, k=min(k, n))
        monkeypatch.setenv("SC_SCAN_QSTREAM", "1")
        d1, r1 = check_exact(ix, X, Q, min(k, n), metric)
        assert np.array_equal(r0, r1) and np.array_equal(bits(d0), bits(d1)), (n, dim, nq, k)
        monkeypatch.delenv("SC_SCAN_QSTREAM")
        ix.close()


def test_streamed_query_variant_is_the_default_for_long_rows(rt):
    """3 072-d rows leave LDS room for 6 resident queries; 16 queries then take the streamed variant in ONE pass."""
    X = orc.synth(6_000, 3072, seed=41)
    Q = orc.synth(16, 3072, seed=42)
    for metric in ("L2", "COSINE"):
        ix = _native.Index(rt, 3072, metric=metric)
        ix.add(X)
        ix.set_search_mode("exact")
        check_exact(ix, X, Q, 10, metric)
        ix.close()


def test_bad_arguments_raise(rt):
    ix = _native.Index(rt, 64, metric="L2")
    ix.add(np.zeros((4, 64), np.float32))
    with pytest.raises(_native.ScError):
        ix.search(np.zeros((1, 64), np.float32), k=0)
    with pytest.raises(_native.ScError):
        ix.search(np.zeros((1, 64), np.float32), k=5000)
    with pytest.raises(ValueError):
        ix.search(np.zeros((1, 63), np.float32), k=1)
    with pytest.raises(_native.ScError):
        ix.overwrite(np.zeros((1, 64), np.float32), [4])
    ix.close()


def test_add_in_batches_overwrite_and_readback(rt):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((5000, 96)).astype(np.float32)
    Q = rng.standard_normal((6, 96)).astype(np.float32)
    ix = _native.Index(rt, 96, metric="COSINE")
    for s in range(0, 5000, 128):  # reference upsert batch size (settings.py:76)
        ix.add(X[s:s + 128])
    assert len(ix) == 5000
    assert np.array_equal(ix.get_rows(0, 5000), X)
    check_exact(ix, X, Q, 10, "COSINE")
    rows = np.array([0, 4999, 77, 1234])
    X[rows] = rng.standard_normal((4, 96)).astype(np.float32)
    ix.overwrite(X[rows], rows)  # upsert of existing primary keys
    assert np.array_equal(ix.get_rows(77, 1), X[77:78])
    check_exact(ix, X, Q, 10, "COSINE")
    ix.close()


def test_duplicate_rows_tie_break(rt):
    rng = np.random.default_rng(4)
    X = rng.standard_normal((3000, 64)).astype(np.float32)
    X[2000:2040] = X[5]  # 41 identical rows spread over many waves' tiles
    q = X[5:6] + 0.001
    for metric in METRICS:
        ix = _native.Index(rt, 64, metric=metric)
        ix.add(X)
        d, r = ix.search(q, k=41)
        assert r[0].tolist() == [5] + list(range(2000, 2040))
        assert len(set(bits(d[0]).tolist())) == 1
        ix.close()


def test_synthetic_fill_matches_oracle_generator(rt):
    ix = _native.Index(rt, 768, metric="L2")
    ix.fill_synthetic(3000, seed=9, first_row=123456)
    got = ix.get_rows(0, 3000)
    assert np.array_equal(bits(got), bits(orc.synth(3000, 768, seed=9, first_row=123456)))
    Q = orc.synth(2, 768, seed=10)
    check_exact(ix, got, Q, 10, "L2")
    ix.close()


def test_full_size_10m_x_768_properties(rt):
    """BASELINE config 3 size.  The oracle cannot scan 30 GB, so check size-independent properties."""
    N, D, SEED, K = 10_000_000, 768, 0, 10
    ix = _native.Index(rt, D, metric="L2")
    ix.fill_synthetic(N, seed=SEED)
    rng = np.random.default_rng(1)
    planted = rng.integers(0, N, size=8)
    Q = orc.synth(16, D, seed=1)
    Q[:8] = orc.synth_rows(planted, D, SEED) + 0.05 * rng.standard_normal((8, D)).astype(np.float32)
    d, r = ix.search(Q, k=K)
    # 1. sorted best-first, ids unique and in range
    assert (np.diff(d, axis=1) >= 0).all() and ((r >= 0) & (r < N)).all()
    assert all(len(set(row)) == K for row in r.tolist())
    # 2. planted neighbours are found first
    assert np.array_equal(r[:8, 0], planted)
    # 3. every returned distance is bit-exact against the oracle on the regenerated rows
    for qi in range(16):
        Xr = orc.synth_rows(r[qi], D, SEED)
        od, orow = orc.search(Xr, Q[qi:qi + 1], K, "L2")
        assert np.array_equal(bits(od[0]), bits(d[qi])) and np.array_equal(orow[0], np.arange(K))
    # 4. no row of a 200k random sample beats the k-th result unless it was returned
    sample = np.unique(rng.integers(0, N, size=200_000))
    Xs = orc.synth_rows(sample, D, SEED)
    sd, sr = orc.search(Xs, Q, K, "L2")
    for qi in range(16):
        better = sample[sr[qi][sd[qi] < d[qi, -1]]]
        assert set(better.tolist()) <= set(r[qi].tolist())
    # 5. shard invariance: top-k(whole) == merge(top-k(halves)), through the host merge entry point
    parts_d, parts_r = [], []
    for base in (0, N // 2):
        sh = _native.Index(rt, D, metric="L2", row_base=base)
        sh.fill_synthetic(N // 2, seed=SEED, first_row=base)
        dd, rr = sh.search(Q, k=K)
        parts_d.append(dd)
        parts_r.append(rr)
        sh.close()
    md, mr = _native.topk_merge_host("L2", np.stack(parts_d), np.stack(parts_r))
    assert np.array_equal(mr, r) and np.array_equal(bits(md), bits(d))
    ix.close()
