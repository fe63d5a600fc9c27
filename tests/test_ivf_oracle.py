"""CPU: the IVF restatement is deterministic, partitions the rows, and its recall behaves as IVF_FLAT should."""
import numpy as np

from oracle import sc_oracle as orc
from oracle.ivf_oracle import IvfOracle, centroid_mean


def test_centroid_mean_is_sequential_f32_mean():
    rng = np.random.default_rng(0)
    S = rng.standard_normal((200, 8)).astype(np.float32)
    assign = rng.integers(0, 5, size=200)
    assign[assign == 3] = 2  # cluster 3 stays empty -> keeps its old centroid
    old = rng.standard_normal((5, 8)).astype(np.float32)
    got = centroid_mean(S, assign, 5, old)
    for c in range(5):
        rows = np.nonzero(assign == c)[0]
        if len(rows) == 0:
            assert np.array_equal(got[c], old[c])
            continue
        acc = np.zeros(8, np.float32)
        for r in rows:
            acc = (acc + S[r]).astype(np.float32)
        assert np.array_equal(got[c], (acc / np.float32(len(rows))).astype(np.float32))


def test_lists_partition_rows_and_recall_grows_with_nprobe():
    rng = np.random.default_rng(1)
    centers = rng.standard_normal((30, 32)).astype(np.float32)
    X = (centers[rng.integers(0, 30, 6000)] + 0.3 * rng.standard_normal((6000, 32))).astype(np.float32)
    Q = (centers[rng.integers(0, 30, 40)] + 0.3 * rng.standard_normal((40, 32))).astype(np.float32)
    ivf = IvfOracle(X, "L2", nlist=24, niter=5)
    again = IvfOracle(X, "L2", nlist=24, niter=5)
    assert np.array_equal(ivf.centroids, again.centroids)
    assert sorted(np.concatenate(ivf.lists).tolist()) == list(range(6000))
    _, exact = orc.search(X, Q, 10, "L2")
    recalls = []
    for nprobe in (1, 4, 24):
        _, r = ivf.search(Q, 10, nprobe)
        recalls.append(np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(r.tolist(), exact.tolist())]))
    assert recalls[0] <= recalls[1] <= recalls[2] and recalls[2] == 1.0 and recalls[1] > 0.8


def test_ivf_restatement_matches_committed_build(golden):
    """tests/golden/ivf_4096x64.npz locks the deterministic build rule (generator: oracle/gen_fixtures.py gen_ivf): same centroids
    bit for bit, same lists, same probe results -- a change of sampling, initialisation, re-seeding or probe order shows up here
    and in tests/test_ivf_gpu.py::test_gpu_build_matches_committed_build."""
    d = np.load(golden / "knn_4096x64.npz")
    g = np.load(golden / "ivf_4096x64.npz")
    for metric in ("IP", "L2", "COSINE"):
        o = IvfOracle(d["X"], metric, nlist=16, niter=6)
        assert np.array_equal(o.centroids.view(np.uint32), g[f"{metric}_centroids"].view(np.uint32))
        assert np.array_equal(o.assign, g[f"{metric}_assign"])
        dist, rows = o.search(d["Q"], 10, 4)
        assert np.array_equal(rows, g[f"{metric}_rows"]) and np.array_equal(dist.view(np.uint32), g[f"{metric}_dist"].view(np.uint32))
