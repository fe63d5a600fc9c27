"""GPU: a random sequence of collection operations -- upserts of new and of known primary keys (also repeated inside one batch),
single and batched searches, an explicit index build, a save / load in the middle -- against a host model of the collection
(dict of primary key -> row, f32 matrix) searched with the oracle.

FLAT collections: ids and f32 distances bit-exact after every operation.  IVF_FLAT collections (the reference's index,
milvus_store.py:76-84): the probe is approximate by design, so each answer is checked for what holds regardless of the lists --
every reported distance is the oracle's distance of that row, best first, no row twice, the primary keys / texts are the row's
current ones -- and the queries that ARE stored rows find themselves (a row sits in the list of its nearest centroid, which is the
first list its own probe visits)."""
import numpy as np
import pytest

from oracle import sc_oracle as orc
from semcode_amd import _native
from semcode_amd.embeddings import EmbeddingPayload
from semcode_amd.storage import MilvusVectorStore

pytestmark = pytest.mark.gpu
DIM = 64


@pytest.fixture(scope="module")
def rt():
    r = _native.Runtime(device=0)
    yield r
    r.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


class Model:
    def __init__(self):
        self.row_of, self.ids, self.texts, self.X = {}, [], [], np.zeros((0, DIM), np.float32)

    def upsert(self, payloads):
        for p in payloads:  # (in order: a key repeated in one call ends with its last vector)
            v = np.asarray(p.vector, np.float32)
            if p.id in self.row_of:
                r = self.row_of[p.id]
                self.X[r] = v
                self.texts[r] = p.text
            else:
                self.row_of[p.id] = len(self.ids)
                self.ids.append(p.id)
                self.texts.append(p.text)
                self.X = np.vstack([self.X, v[None]])


def payload(rng, key, gen):
    v = rng.standard_normal(DIM).astype(np.float32)
    return EmbeddingPayload(id=key, text=f"{key}@{gen}", vector=v.tolist(),
                            metadata={"repo": "r", "path": key, "language": "py", "start_line": 1, "end_line": 2, "symbol": None})


@pytest.mark.parametrize("metric", ["L2", "IP", "COSINE"])
@pytest.mark.parametrize("index_type", ["FLAT", "IVF_FLAT"])
def test_random_collection_operations(rt, tmp_path, metric, index_type):
    rng = np.random.default_rng({"L2": 1, "IP": 2, "COSINE": 3}[metric] * 10 + len(index_type))
    store = MilvusVectorStore(dim=DIM, metric=metric, index_type=index_type, nlist=16, nprobe=4, runtime=rt)
    store.connect()
    model = Model()
    next_key = 0
    searches = 0
    for step in range(110):
        op = str(rng.choice(["new", "new", "mixed", "known", "search", "search", "build", "saveload"]))
        if not model.ids:
            op = "new"
        if op == "new":
            n = int(rng.choice([1, 5, 130, 700, 3000])) if model.ids else 130
            batch = [payload(rng, f"k{next_key + i}", step) for i in range(n)]
            next_key += n
        elif op == "known":
            batch = [payload(rng, model.ids[int(j)], step) for j in rng.integers(0, len(model.ids), size=int(rng.choice([1, 40])))]  # (keys may repeat)
        elif op == "mixed":
            batch = [payload(rng, model.ids[int(j)], step) for j in rng.integers(0, len(model.ids), size=20)]
            batch += [payload(rng, f"k{next_key + i}", step) for i in range(30)]
            next_key += 30
            batch = [batch[int(j)] for j in rng.permutation(len(batch))]
        if op in ("new", "known", "mixed"):
            store.upsert_embeddings(batch)
            model.upsert(batch)
            assert len(store) == len(model.ids)
            continue
        if op == "build":
            store.build_index(niter=3)
            continue
        if op == "saveload":
            store.save(tmp_path / f"c{step}")
            store.close()
            store = MilvusVectorStore(dim=DIM, metric=metric, index_type=index_type, nlist=16, nprobe=4, runtime=rt)
            store.connect()
            store.load(tmp_path / f"c{step}")
            assert len(store) == len(model.ids)
            continue
        # search: some queries are stored rows, the rest random; once as a batch, one of them again through search()
        searches += 1
        n = len(model.ids)
        Q = int(rng.choice([1, 3, 20, 70, 300]))
        k = int(rng.choice([1, 5, 10, 70]))
        own = rng.integers(0, n, size=max(1, Q // 2))
        q = rng.standard_normal((Q, DIM)).astype(np.float32)
        q[: len(own)] = model.X[own]
        dist, rows = store.search_batch(q, top_k=k)
        want_d, want_r = orc.search(model.X, q, min(k, n), metric)
        kk = min(k, n)
        if index_type == "FLAT":
            assert np.array_equal(rows[:, :kk], want_r) and np.array_equal(bits(dist[:, :kk]), bits(want_d)), (step, n, Q, k)
        else:
            for j in range(Q):
                got = rows[j][rows[j] >= 0]
                assert len(set(got.tolist())) == len(got) and len(got) >= 1
                d1, r1 = orc.search_rows(model.X, q[j], got, len(got), metric)  # the oracle's order and distances over exactly these rows
                assert np.array_equal(r1, got) and np.array_equal(bits(d1), bits(dist[j][: len(got)])), (step, j)
            if metric != "IP":  # (under IP a row need not be its own best match)
                for j in range(len(own)):
                    assert np.array_equal(model.X[rows[j, 0]], model.X[own[j]]), (step, j)
        hits = next(iter(store.search(q[0].tolist(), top_k=k)))  # (a batch of one may take another path than the batch above: its own reference)
        d0, r0 = store.search_batch(q[:1], top_k=k)
        assert [h.id for h in hits] == [model.ids[int(r)] for r in r0[0] if r >= 0]
        assert [h.entity.get("text") for h in hits] == [model.texts[int(r)] for r in r0[0] if r >= 0]
        assert np.array_equal(bits([h.distance for h in hits]), bits(d0[0][r0[0] >= 0]))
        if index_type == "FLAT":
            assert np.array_equal(r0[0], rows[0])
    assert searches >= 5
    store.close()
