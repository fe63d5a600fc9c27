"""GPU: the batched search path (int8 / bf16 MFMA coarse scan + exact f32 re-rank + certificate) returns exactly what
the exact scan and the CPU oracle return: ids/order exact, distances bit-exact -- whichever coarse stage answers
(stage 0 = int8 first, then bf16, then the exact scan; 8 / 16 = that stage pinned)."""
import numpy as np
import pytest

from oracle import sc_oracle as orc
from semcode_amd import _native

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", ["IP", "L2", "COSINE"])
@pytest.mark.parametrize("nq", [17, 64, 200])
def test_batched_equals_oracle_100k(rt, metric, nq):
    X = orc.synth(100_000, 768, seed=31)
    Q = orc.synth(nq, 768, seed=32)
    od, orow = orc.search(X, Q, 10, metric)
    ix = _native.Index(rt, 768, metric=metric)
    ix.add(X)
    ix.set_search_mode("batched")
    for stage in (0, 8, 16):
        ix.set_coarse_stage(stage)
        d, r = ix.search(Q, k=10)
        st = ix.last_search_stats()
        assert st["path"] == "batched" and st["coarse_bits"] == (16 if stage == 16 else 8), st
        assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)), (stage, st)
        # the certificates must hold for the bulk of gaussian queries, at either precision
        assert st["uncertified"] <= nq // 4 and st["handed_to_bf16"] <= nq // 4, st
    ix.close()


@pytest.mark.parametrize("dim,metric", [(768, "L2"), (768, "IP"), (768, "COSINE"), (448, "L2"), (192, "L2")])
def test_persistent_coarse_kernel_walks_many_tiles_per_workgroup(rt, dim, metric):
    """The coarse scan is a persistent kernel whose LDS ring prefetches the next tile's first two K-tiles (scan_coarse256p_kernel).
    With the grid cut to 8 / 24 workgroups each one walks dozens of tiles -- even and odd K-tile counts (the ring parity flips
    between tiles when odd: 448 dims = 7 bf16 K-tiles), both coarse stages -- and with one workgroup per tile (the non-persistent
    kernel) the results are the same bits."""
    # (int8 stage: the per-wave hit lists of the persistent kernel exist from 256 x 512 = 131 072 rows up; bf16: from 32 768)
    X = orc.synth(140_000 if (dim, metric) == (768, "L2") else 60_000, dim, seed=41)
    Q = orc.synth(300, dim, seed=42)  # two query tiles
    od, orow = orc.search(X, Q, 10, metric)
    ix = _native.Index(rt, dim, metric=metric)
    ix.add(X)
    ix.set_search_mode("batched")
    try:
        for wgs, persistent in ((8, 1), (24, 1), (0, 1), (0, 0)):
            _native.diag_set_option("coarse_workgroups", wgs)
            _native.diag_set_option("coarse_persistent", persistent)
            for stage in (8, 16):
                ix.set_coarse_stage(stage)
                d, r = ix.search(Q, k=10)
                assert ix.last_search_stats()["path"] == "batched"
                assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)), (wgs, persistent, stage)
    finally:
        _native.diag_set_option("coarse_workgroups", 0)
        _native.diag_set_option("coarse_persistent", 1)
        ix.close()


@pytest.mark.parametrize("dim,metric,nq", [(768, "L2", 17), (768, "IP", 40), (768, "COSINE", 64), (192, "L2", 33), (100, "L2", 64), (448, "IP", 20)])
def test_small_batches_take_the_narrow_streaming_kernel(rt, dim, metric, nq):
    """17 .. 64 queries on the int8 stage: the sparse phases (from 256 x 512 rows on) run scan_coarse64s_kernel -- 256 rows x 64
    query slots per tile, the shadow as one stream of stages through a three-deep ring, survivors appended to per-wave lists and
    scattered afterwards.  Several stage counts per tile (768 dims = 6, 448 = 4, 192 = 2, 100 = 1), a ragged last tile, few
    workgroups walking many tiles; the 256-query tiles (SC_COARSE64=0 is the A/B switch) and the oracle give the same bits."""
    n = 200_000 + 77
    X = orc.synth(n, dim, seed=51)
    Q = orc.synth(nq, dim, seed=52)
    od, orow = orc.search(X, Q, 10, metric)
    ix = _native.Index(rt, dim, metric=metric)
    ix.add(X)
    ix.set_search_mode("batched")
    ix.set_coarse_stage(8)
    try:
        for wgs in (0, 16):
            _native.diag_set_option("coarse_workgroups", wgs)
            d, r = ix.search(Q, k=10)
            st = ix.last_search_stats()
            assert st["path"] == "batched" and st["coarse_bits"] == 8
            assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)), (wgs, st)
    finally:
        _native.diag_set_option("coarse_workgroups", 0)
        ix.close()


@pytest.mark.parametrize("n", [1, 100, 127, 128, 129, 1000, 5000])
def test_batched_small_and_ragged(rt, n):
    rng = np.random.default_rng(n)
    for dim in (100, 192):  # row strides of 128 and 192 floats: the int8 rows are padded to whole 128-byte K-tiles
        X = rng.standard_normal((n, dim)).astype(np.float32)
        Q = rng.standard_normal((40, dim)).astype(np.float32)
        od, orow = orc.search(X, Q, 7, "L2", row_base=77)
        ix = _native.Index(rt, dim, metric="L2", row_base=77)
        ix.add(X)
        ix.set_search_mode("batched")
        for stage in (8, 16):
            ix.set_coarse_stage(stage)
            d, r = ix.search(Q, k=7)
            assert ix.last_search_stats()["path"] == "batched"
            assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)), (dim, stage)
        ix.close()


def test_int8_stage_hands_hard_queries_on_and_switches_itself_off(rt):
    """A corpus the int8 stage cannot separate (outlier dimensions blow up the per-row quantisation step): its uncertified
    queries are answered by the bf16 stage, results stay exact, and the index starts later searches at the bf16 stage."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((50_000, 256)).astype(np.float32)
    X[:, :4] = 300.0   # 4 constant outlier dimensions: every row's step is 300 / 127, the 252 informative dimensions round to 0 or +-1
    Q = rng.standard_normal((64, 256)).astype(np.float32)
    Q[:, :4] = 0.0     # neighbours are decided by the small dimensions, which bf16 keeps to 8 bits each
    od, orow = orc.search(X, Q, 10, "L2")
    ix = _native.Index(rt, 256, metric="L2")
    ix.add(X)
    ix.set_search_mode("batched")
    d, r = ix.search(Q, k=10)
    st = ix.last_search_stats()
    assert st["coarse_bits"] == 8 and st["handed_to_bf16"] > 16, st
    assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
    d, r = ix.search(Q, k=10)
    st = ix.last_search_stats()
    assert st["coarse_bits"] == 16 and st["handed_to_bf16"] == 0, st
    assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
    ix.set_coarse_stage(0)  # clears the switch
    ix.search(Q, k=10)
    assert ix.last_search_stats()["coarse_bits"] == 8
    ix.close()


def test_batched_incremental_adds_and_overwrite(rt):
    rng = np.random.default_rng(5)
    X = rng.standard_normal((6000, 128)).astype(np.float32)
    Q = rng.standard_normal((33, 128)).astype(np.float32)
    ix = _native.Index(rt, 128, metric="IP")
    ix.set_search_mode("batched")
    ix.add(X[:3000])
    d, r = ix.search(Q, k=5)
    assert np.array_equal(r, orc.search(X[:3000], Q, 5, "IP")[1])
    ix.add(X[3000:])  # shadow must be extended
    d, r = ix.search(Q, k=5)
    assert np.array_equal(r, orc.search(X, Q, 5, "IP")[1])
    X[[10, 5999]] = 3.0 * Q[[0, 1]]  # overwrite -> shadow rebuilt; these rows become the best hits
    ix.overwrite(X[[10, 5999]], [10, 5999])
    d, r = ix.search(Q, k=5)
    od, orow = orc.search(X, Q, 5, "IP")
    assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)) and r[0, 0] == 10 and r[1, 0] == 5999
    ix.close()


@pytest.mark.parametrize("metric", ["IP", "L2", "COSINE"])
def test_near_duplicates_defeat_the_certificate_but_not_the_result(rt, metric):
    # 600 rows that differ from one base vector by ~1e-4: bf16 cannot separate them, so the certificate fails
    # and the query is re-run exactly; exact duplicates must still come back in row order.
    rng = np.random.default_rng(9)
    X = rng.standard_normal((8000, 256)).astype(np.float32)
    base = rng.standard_normal(256).astype(np.float32)
    X[1000:1600] = base + 1e-4 * rng.standard_normal((600, 256)).astype(np.float32)
    X[1200:1210] = base
    Q = rng.standard_normal((20, 256)).astype(np.float32)
    Q[0] = base
    ix = _native.Index(rt, 256, metric=metric)
    ix.add(X)
    ix.set_search_mode("batched")
    od, orow = orc.search(X, Q, 16, metric)
    try:
        # the collect pass (every row within the coarse error of the k-th exact score, re-scored exactly) answers the query ...
        d, r = ix.search(Q, k=16)
        st = ix.last_search_stats()
        assert st["path"] == "batched" and st["collect_resolved"] >= 1 and st["uncertified"] == 0, st
        assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
        # ... and without it the exact scan does
        _native.diag_set_option("collect_pass", 0)
        d, r = ix.search(Q, k=16)
        st = ix.last_search_stats()
        assert st["path"] == "batched" and st["uncertified"] >= 1 and st["collect_tried"] == 0, st
        assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
    finally:
        _native.diag_set_option("collect_pass", 1)
    ix.set_search_mode("exact")
    d2, r2 = ix.search(Q, k=16)
    assert np.array_equal(r2, r) and np.array_equal(bits(d2), bits(d))
    ix.close()


@pytest.mark.parametrize("metric", ["IP", "L2", "COSINE"])
@pytest.mark.parametrize("stage", [8, 16])
def test_collect_pass_answers_clustered_batches_and_hands_on_what_it_cannot_hold(rt, metric, stage):
    """Tight clusters: hundreds of rows lie within the coarse error of the k-th neighbour, far more than the candidates a stage keeps, so
    its certificate fails for most of the batch.  The collect pass re-scores every row within the bound and answers them; a query
    sitting on 5000 near-duplicates (more than the pass keeps per query) goes on to the next stage.  Bit-exact throughout."""
    rng = np.random.default_rng(21)
    ncl, per, dim = 12, 1500, 192
    centers = 4.0 * rng.standard_normal((ncl, dim)).astype(np.float32)
    X = (np.repeat(centers, per, axis=0) + 0.02 * rng.standard_normal((ncl * per, dim))).astype(np.float32)
    X = np.concatenate([X, (X[7][None, :] + 1e-5 * rng.standard_normal((5000, dim))).astype(np.float32)])
    Q = (centers[rng.integers(0, ncl, size=47)] + 0.02 * rng.standard_normal((47, dim))).astype(np.float32)
    Q = np.concatenate([Q, X[7:8] + 1e-6])
    od, orow = orc.search(X, Q, 10, metric)
    ix = _native.Index(rt, dim, metric=metric)
    ix.add(X)
    ix.set_search_mode("batched")
    ix.set_coarse_stage(stage)
    d, r = ix.search(Q, k=10)
    st = ix.last_search_stats()
    assert st["path"] == "batched" and st["coarse_bits"] == stage, st
    assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od)), st
    assert st["collect_tried"] >= 24 and st["collect_resolved"] >= st["collect_tried"] - 8, st  # the batch needed it, and it answered
    assert st["collect_resolved"] < st["collect_tried"] or st["uncertified"] == 0, st
    ix.close()


def test_auto_mode_picks_paths(rt):
    X = orc.synth(20_000, 128, seed=3)
    ix = _native.Index(rt, 128, metric="L2")
    ix.add(X)
    ix.search(orc.synth(8, 128, seed=4), k=10)
    assert ix.last_search_stats()["path"] == "exact"
    ix.search(orc.synth(100, 128, seed=4), k=10)
    assert ix.last_search_stats()["path"] == "batched"
    ix.search(orc.synth(100, 128, seed=4), k=100)  # k too large for the 128-candidate re-rank
    assert ix.last_search_stats()["path"] == "exact"
    ix.close()


def test_full_size_batched_1024_queries_over_10m(rt):
    """BASELINE config 3.  Batched result == exact-scan result on a sample of the queries; properties on all."""
    N, D, K = 10_000_000, 768, 10
    ix = _native.Index(rt, D, metric="L2")
    ix.fill_synthetic(N, seed=0)
    Q = orc.synth(1024, D, seed=1)
    d, r = ix.search(Q, k=K)
    st = ix.last_search_stats()
    assert st["path"] == "batched" and st["coarse_bits"] == 8 and st["uncertified"] <= 8 and st["handed_to_bf16"] <= 32, st
    ix.set_coarse_stage(16)  # the bf16 stage alone returns the very same thing
    d16, r16 = ix.search(Q, k=K)
    st16 = ix.last_search_stats()
    assert st16["coarse_bits"] == 16 and st16["uncertified"] <= 8, st16
    assert np.array_equal(r16, r) and np.array_equal(bits(d16), bits(d))
    ix.set_coarse_stage(0)
    assert (np.diff(d, axis=1) >= 0).all() and ((r >= 0) & (r < N)).all()
    # returned distances are bit-exact f32 scores of the regenerated rows
    for qi in range(0, 1024, 97):
        od, orow = orc.search(orc.synth_rows(r[qi], D, 0), Q[qi:qi + 1], K, "L2")
        assert np.array_equal(bits(od[0]), bits(d[qi])) and np.array_equal(orow[0], np.arange(K))
    # and equal to the exact HBM-bound scan on 32 of the queries
    ix.set_search_mode("exact")
    sel = np.arange(0, 1024, 32)
    de, re_ = ix.search(Q[sel], k=K)
    assert np.array_equal(re_, r[sel]) and np.array_equal(bits(de), bits(d[sel]))
    ix.close()


@pytest.mark.parametrize("metric", ["L2", "IP", "COSINE"])
def test_wide_candidate_set_answers_clustered_batches_in_one_pass(rt, metric):
    """The int8 stage in its wide form (every key within the exact-score cut kept between the phases, the certificate by
    construction): a clustered corpus of 400k rows whose clusters hold more rows than the 512 candidates of the plain form; same bits
    as the exact scan, no query left over, no collect pass."""
    n, dim, ncl = 400_000, 64, 300
    ix = _native.Index(rt, dim, metric=metric)
    ix.fill_synthetic_clustered(n, seed=3, nclusters=ncl, spread=0.05)
    qs = _native.Index(rt, dim, metric=metric)
    qs.fill_synthetic_clustered(96, seed=3, nclusters=ncl, spread=0.05, first_row=n + 999)
    Q = qs.get_rows(0, 96)
    qs.close()
    try:
        ix.set_search_mode("exact")
        d0, r0 = ix.search(Q, k=10)
        ix.set_search_mode("batched")
        ix.set_coarse_stage(8)
        _native.diag_set_option("wide_candidates", 1)
        d1, r1 = ix.search(Q, k=10)
        st = ix.last_search_stats()
        assert st["path"] == "batched" and st["coarse_bits"] == 8 and st["wide"], st
        assert np.array_equal(r0, r1) and np.array_equal(bits(d0), bits(d1)), st
        assert st["uncertified"] == 0 and st["collect_tried"] == 0, st
        _native.diag_set_option("wide_candidates", 0)
        d2, r2 = ix.search(Q, k=10)  # the plain form: its certificate fails, the collect pass (or the exact scan) answers
        st2 = ix.last_search_stats()
        assert not st2["wide"] and st2["collect_tried"] + st2["uncertified"] >= 48, st2
        assert np.array_equal(r0, r2) and np.array_equal(bits(d0), bits(d2))
    finally:
        _native.diag_set_option("wide_candidates", 0)
        ix.close()


@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_top_k_beyond_64_takes_the_wide_form_of_the_int8_stage(rt, metric):
    """top_k 65 .. 128 used to fall to the exact scan (one pass per 16 queries); the int8 stage's wide form (512 candidates re-scored for
    the cut, every key within it kept) answers it with the same bits.  top_k beyond 128 still takes the exact scan."""
    n, dim = 300_000, 64
    ix = _native.Index(rt, dim, metric=metric)
    ix.fill_synthetic(n, seed=11)
    Q = orc.synth(40, dim, seed=12)
    ix.set_search_mode("exact")
    d0, r0 = ix.search(Q, k=100)
    ix.set_search_mode("batched")
    d1, r1 = ix.search(Q, k=100)
    st = ix.last_search_stats()
    assert st["path"] == "batched" and st["coarse_bits"] == 8 and st["wide"], st
    assert np.array_equal(r0, r1) and np.array_equal(bits(d0), bits(d1)), st
    d2, r2 = ix.search(Q, k=200)
    assert ix.last_search_stats()["path"] == "exact"
    d3, r3 = ix.search(Q[:5], k=10)  # the large top_k left no switch behind
    assert ix.last_search_stats()["path"] == "batched" and not ix.last_search_stats()["wide"]
    ix.close()
