"""CPU restatement of the IVF_FLAT build and probe search.  TEST INFRASTRUCTURE ONLY.

Parity status: "parity unpinned" -- the reference's IVF_FLAT lives in the Milvus 2.4.4 server
(src/semcode/storage/milvus_store.py:76-84,141-147; docker-compose.yml:5), whose k-means uses random
sampling/initialisation and is absent offline.  This restates the DETERMINISTIC build of
semcode_amd/csrc/sc_ivf.cpp (sampling, initialisation, Lloyd iterations, assignment metric, probe
rule), using the canonical scoring of sc_oracle.c, so that centroids, lists and results can be compared
exactly; recall against the exhaustive oracle is the size-independent property used at full scale.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import sc_oracle as orc


def _assign_metric(metric: str) -> str:
    return "COSINE" if metric == "COSINE" else "L2"


def _nearest(Cent: np.ndarray, X: np.ndarray, metric: str) -> np.ndarray:
    _, rows = orc.search(Cent, X, 1, metric)
    return rows[:, 0].astype(np.int64)


def centroid_mean(S: np.ndarray, assign: np.ndarray, nlist: int, C_old: np.ndarray) -> np.ndarray:
    S = np.ascontiguousarray(S, dtype=np.float32)
    order = np.argsort(assign, kind="stable").astype(np.int64)
    off = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(np.bincount(assign, minlength=nlist), out=off[1:])
    C_old = np.ascontiguousarray(C_old, dtype=np.float32)
    out = np.empty((nlist, S.shape[1]), dtype=np.float32)
    orc.lib().sc_oracle_centroid_mean(S.ctypes.data_as(C.c_void_p), S.shape[1], order.ctypes.data_as(C.c_void_p),
                                      off.ctypes.data_as(C.c_void_p), nlist, C_old.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def reseed(Cent: np.ndarray, cnt: np.ndarray, ns: int) -> int:
    """Between Lloyd iterations (sc_ivf.cpp sc_index_train): starved centroids (count < 0.75 average, smallest first, ties by
    id) are moved onto the largest ones (count > 2 average, largest first, ties by lower id; the size is halved on every
    split), the two copies pushed apart by the factors (1 +- 2^-10), sign alternating over the dimensions.  In place."""
    import heapq

    nlist, dim = Cent.shape
    cnt = [int(c) for c in cnt]
    donors = sorted(range(nlist), key=lambda c: cnt[c])  # stable: ties keep ascending id
    heap = [(-cnt[c], c) for c in range(nlist) if cnt[c] * nlist > 2 * ns]
    heapq.heapify(heap)
    sg = np.where(np.arange(dim) % 2 == 1, -1.0, 1.0).astype(np.float32) * np.float32(1.0 / 1024.0)
    up, down = (np.float32(1.0) + sg).astype(np.float32), (np.float32(1.0) - sg).astype(np.float32)
    moves = 0
    for e in donors:
        if not heap or cnt[e] * 4 * nlist >= 3 * ns:
            break
        size, b = heapq.heappop(heap)
        size = -size
        if size * nlist <= 2 * ns:
            break
        v = Cent[b].copy()
        Cent[e] = v * up
        Cent[b] = v * down
        half = size // 2
        heapq.heappush(heap, (-(size - half), b))
        heapq.heappush(heap, (-half, e))
        moves += 1
    return moves


class IvfOracle:
    def __init__(self, X: np.ndarray, metric: str, nlist: int, niter: int = 10):
        X = np.ascontiguousarray(X, dtype=np.float32)
        n = X.shape[0]
        self.X, self.metric = X, metric
        self.nlist = nlist = min(nlist, n)
        ns = min(n, 256 * nlist)
        srows = (np.arange(ns, dtype=object) * n // ns).astype(np.int64)
        S = X[srows]
        crow = (np.arange(nlist, dtype=object) * ns // nlist).astype(np.int64)
        Cent = S[crow].copy()
        am = _assign_metric(metric)
        for it in range(niter):
            assign = _nearest(Cent, S, am)
            Cent = centroid_mean(S, assign, nlist, Cent)
            if it + 1 < niter:
                reseed(Cent, np.bincount(assign, minlength=nlist), ns)
        self.centroids = Cent
        self.assign = _nearest(Cent, X, am)
        self.lists = [np.nonzero(self.assign == c)[0] for c in range(nlist)]

    def search(self, Q: np.ndarray, k: int, nprobe: int):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        _, probe = orc.search(self.centroids, Q, nprobe, self.metric)  # probe under the index metric
        dist = np.empty((len(Q), k), np.float32)
        rows = np.empty((len(Q), k), np.int64)
        for i in range(len(Q)):
            cand = np.concatenate([self.lists[c] for c in probe[i] if c >= 0]) if nprobe else np.zeros(0, np.int64)
            dist[i], rows[i] = orc.search_rows(self.X, Q[i], np.sort(cand), k, self.metric)
        return dist, rows
