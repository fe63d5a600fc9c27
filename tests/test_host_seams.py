"""CPU: host logic of the two seams (progress protocols, upsert-by-primary-key, hit shape, errors).

The device index is replaced by a tiny in-memory stand-in that records calls; the numerical path is
covered by the -m gpu tests.  Protocol pins come from SURVEY.md section 8a ("Caller pins"), read from
reference src/semcode/storage/milvus_store.py:87-148, src/semcode/services/indexer.py:135-188 and
src/semcode/rag/pipeline.py:112-169.
"""
import json
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import pytest

from semcode_amd.embeddings.payload import EmbeddingPayload
from semcode_amd.services import build_payloads, ingest_chunks, make_chunk_id
from semcode_amd.settings import settings
from semcode_amd.storage import MilvusVectorStore


class FakeIndex:
    """Host stand-in with the _native.Index surface (add / overwrite / search); IP metric, exact."""

    def __init__(self, dim, **_):
        self.dim = dim
        self.X = np.zeros((0, dim), np.float32)
        self.calls = []

    def add(self, v):
        self.calls.append(("add", len(v)))
        self.X = np.concatenate([self.X, np.asarray(v, np.float32)])

    def overwrite(self, v, rows):
        self.calls.append(("overwrite", [int(r) for r in rows]))
        self.X[np.asarray(rows)] = v

    def put_rows(self, v, rows):
        rows = [int(r) for r in rows]
        self.calls.append(("put_rows", rows))
        v = np.asarray(v, np.float32)
        for vec, r in zip(v, rows):
            if r == len(self.X):
                self.X = np.concatenate([self.X, vec[None]])
            else:
                assert 0 <= r < len(self.X), r
                self.X[r] = vec

    def get_rows(self, first, n):
        return self.X[first:first + n].copy()

    def __len__(self):
        return len(self.X)

    def search(self, q, k=10, nprobe=16):
        s = q @ self.X.T
        order = np.argsort(-s, axis=1, kind="stable")[:, :k]
        rows = np.full((len(q), k), -1, np.int64)
        dist = np.full((len(q), k), -np.inf, np.float32)
        rows[:, : order.shape[1]] = order
        dist[:, : order.shape[1]] = np.take_along_axis(s, order, 1)
        return dist, rows


def make_store(dim=4):
    return MilvusVectorStore(dim=dim, index_factory=lambda **kw: FakeIndex(kw["dim"]))


def payload(i, vec, repo="demo"):
    return EmbeddingPayload(id=f"id{i}", text=f"text {i}", vector=list(vec),
                            metadata={"repo": repo, "path": f"src/f{i}.py", "language": "python", "start_line": 1, "end_line": 2, "symbol": None})


def test_defaults_match_reference():
    s = MilvusVectorStore()
    assert s.collection_name == "semcode_chunks" and s.dim == 3072 and s._collection is None
    assert (s.metric, s.index_type, s.nlist, s.nprobe) == ("IP", "IVF_FLAT", 128, 16)
    assert settings.embedding_batch_size == 64 and settings.milvus_upsert_batch_size == 128 and settings.rag_max_context_sources == 5


def test_use_before_connect_raises_reference_message():
    s = make_store()
    with pytest.raises(RuntimeError, match=r"Milvus collection is not initialized\. Call connect\(\) first\."):
        s.upsert_embeddings([])
    with pytest.raises(RuntimeError, match=r"Milvus collection is not initialized\. Call connect\(\) first\."):
        s.search([0.0] * 4)


def test_upsert_progress_protocol_300_by_128():
    s = make_store()
    s.connect()
    seen = []
    s.upsert_embeddings([payload(i, np.eye(4)[i % 4]) for i in range(300)], progress=lambda a, b: seen.append((a, b)))
    assert seen == [(0, 300), (128, 300), (256, 300), (300, 300)]
    # one native call per reference batch (milvus_store.py:119-130), rows = the next free ones
    assert [c for c in s._collection.calls] == [("put_rows", list(range(0, 128))), ("put_rows", list(range(128, 256))), ("put_rows", list(range(256, 300)))]
    seen.clear()
    s.upsert_embeddings([], progress=lambda a, b: seen.append((a, b)))
    assert seen == [(0, 0)]


def test_upsert_replaces_by_primary_key():
    s = make_store()
    s.connect()
    s.upsert_embeddings([payload(0, [1, 0, 0, 0]), payload(1, [0, 1, 0, 0])])
    s.upsert_embeddings([payload(1, [0, 0, 1, 0]), payload(2, [0, 0, 0, 1]), payload(2, [0, 0, 0, 2])])
    assert len(s) == 3
    assert ("put_rows", [1, 2]) in s._collection.calls  # existing row 1 replaced, row 2 appended, in one call
    assert s._collection.X.tolist() == [[1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 2]]  # last duplicate in a batch wins
    hits = next(iter(s.search([0, 0, 1, 0], top_k=1)))
    assert hits[0].id == "id1" and hits[0].entity.get("path") == "src/f1.py"


def test_search_result_shape_consumed_like_the_pipeline():
    # mirrors SemanticSearchPipeline._retrieve_documents / _hit_to_document (pipeline.py:112-169)
    s = make_store()
    s.connect()
    s.upsert_embeddings([payload(i, v) for i, v in enumerate([[1, 0, 0, 0], [0.9, 0.1, 0, 0], [0, 1, 0, 0]])])
    results = s.search([1, 0, 0, 0], top_k=2)
    assert results
    hits = next(iter(results))
    docs = []
    for hit in hits:
        fetch = hit.entity.get
        score = 0.0
        for attr in ("score", "distance", "similarity"):
            if hasattr(hit, attr):
                score = float(getattr(hit, attr))
                break
        docs.append({"repo": fetch("repo"), "path": fetch("path"), "language": fetch("language"), "snippet": fetch("text") or "",
                     "score": score, "metadata": fetch("metadata") or {}})
    assert [d["path"] for d in docs] == ["src/f0.py", "src/f1.py"]  # best first (IP: descending)
    assert docs[0]["score"] == pytest.approx(1.0) and docs[0]["metadata"]["language"] == "python"
    # fewer rows than top_k: only real hits come back
    assert len(next(iter(s.search([1, 0, 0, 0], top_k=10)))) == 3
    # empty collection: one empty Hits
    e = make_store()
    e.connect()
    assert len(next(iter(e.search([1, 0, 0, 0], top_k=5)))) == 0


def test_dimension_mismatch_is_an_error():
    s = make_store()
    s.connect()
    with pytest.raises(ValueError):
        s.upsert_embeddings([payload(0, [1, 0, 0])])
    with pytest.raises(ValueError):
        s.search([1, 0, 0], top_k=1)


def test_chunk_id_rule(golden):
    for c in json.loads((golden / "chunk_id_kat.json").read_text()):
        assert make_chunk_id(c["repo"], Path(c["path"]), c["start"], c["end"]) == c["md5"]


@dataclass
class Chunk:
    content: str
    path: Path
    language: str
    start_line: int
    end_line: int
    symbol: str | None = None


class DummyEmbedding:  # reference tests/integration/test_indexer_service.py:7-12
    def __init__(self):
        self.batches = []

    def embed_documents(self, texts):
        self.batches.append(len(texts))
        return [[float(len(t))] for t in texts]

    def embed_query(self, text):
        return [float(len(text))]


def test_build_payloads_progress_and_mapping():
    root = Path("/w/demo")
    chunks = [Chunk("x" * (i + 1), root / "src" / f"f{i}.py", "python", 10 * i + 1, 10 * i + 9, None) for i in range(130)]
    emb, seen = DummyEmbedding(), []
    payloads = build_payloads("demo", root, chunks, emb, progress=lambda a, b: seen.append((a, b)))
    assert seen == [(0, 130), (64, 130), (128, 130), (130, 130)] and emb.batches == [64, 64, 2]
    p = payloads[3]
    assert p.id == make_chunk_id("demo", root / "src" / "f3.py", 31, 39)  # absolute workspace path in the id
    assert p.metadata == {"repo": "demo", "path": "src/f3.py", "language": "python", "start_line": 31, "end_line": 39, "symbol": None}
    assert p.vector == [4.0] and p.text == "xxxx"
    emb, seen = DummyEmbedding(), []
    assert build_payloads("demo", root, [], emb, progress=lambda a, b: seen.append((a, b))) == []
    assert seen == [(0, 0)] and emb.batches == []


def test_save_and_load_round_trip(tmp_path):
    s = make_store()
    s.connect()
    s.upsert_embeddings([payload(i, np.eye(4)[i % 4] * (i + 1)) for i in range(10)])
    s.upsert_embeddings([payload(3, [9, 9, 9, 9])])  # an overwritten row must persist as overwritten
    s.save(tmp_path / "col")
    assert sorted(p.name for p in (tmp_path / "col").iterdir()) == ["columns.jsonl", "manifest.json", "vectors.f32"]
    t = make_store()
    t.connect()
    t.load(tmp_path / "col")
    assert len(t) == 10 and np.array_equal(t._collection.X, s._collection.X)
    a, b = next(iter(s.search([1, 1, 1, 1], top_k=4))), next(iter(t.search([1, 1, 1, 1], top_k=4)))
    assert [h.id for h in a] == [h.id for h in b] and a[0].id == "id3"
    assert b[0].entity.get("metadata")["language"] == "python"
    # re-ingest after load is still idempotent by primary key
    t.upsert_embeddings([payload(3, [1, 0, 0, 0])])
    assert len(t) == 10
    wrong = MilvusVectorStore(dim=5, index_factory=lambda **kw: FakeIndex(kw["dim"]))
    wrong.connect()
    with pytest.raises(ValueError):
        wrong.load(tmp_path / "col")


def test_upsert_arrays_equals_upsert_embeddings():
    """The array fast path ends in the same collection as the payload path: same rows, same columns,
    replace-by-primary-key, last duplicate wins, same progress protocol (milvus_store.py:98-133)."""
    rng = np.random.default_rng(0)
    vec = rng.standard_normal((300, 4)).astype(np.float32)
    ids = [f"id{i % 290}" for i in range(300)]  # ids 0..9 occur twice (rows 290..299 replace rows 0..9)
    texts = [f"text {i}" for i in range(300)]
    metas = [{"repo": "demo", "path": f"src/f{i}.py", "language": "python", "start_line": 1, "end_line": 2, "symbol": None} for i in range(300)]
    a, b = make_store(), make_store()
    a.connect(), b.connect()
    a.upsert_embeddings([EmbeddingPayload(id=ids[i], text=texts[i], vector=vec[i].tolist(), metadata=metas[i]) for i in range(300)])
    seen = []
    b.upsert_arrays(ids, vec, texts, metas, progress=lambda x, y: seen.append((x, y)))
    assert seen == [(0, 300), (128, 300), (256, 300), (300, 300)]
    assert len(a) == len(b) == 290 and np.array_equal(a._collection.X, b._collection.X)
    assert a._ids == b._ids and a._texts == b._texts and a._paths == b._paths and a._metadata == b._metadata
    assert b._texts[3] == "text 293" and b._row_of["id3"] == 3
    seen = []
    b.upsert_arrays([], np.zeros((0, 4), np.float32), [], [], progress=lambda x, y: seen.append((x, y)))
    assert seen == [(0, 0)]
    with pytest.raises(ValueError):
        b.upsert_arrays(["x"], np.zeros((1, 5), np.float32), ["t"], [{}])
    with pytest.raises(ValueError):
        b.plan_rows(["p", "p"])
    fresh = make_store()
    with pytest.raises(RuntimeError, match="Call connect"):
        fresh.upsert_arrays(["x"], np.zeros((1, 4), np.float32), ["t"], [{}])


class TokenizingEmbedding:
    """Stand-in for MI355XEmbeddings on the fused path: tokenize() + embed_ids_into(); vector = [number of tokens, first id]."""

    def __init__(self, fail_at=None):
        self.tokenized, self.fail_at = [], fail_at

    def tokenize(self, texts):
        if self.fail_at is not None and len(self.tokenized) == self.fail_at:
            raise RuntimeError("tokenizer broke")
        self.tokenized.append(len(texts))
        lens = np.array([len(t) for t in texts], np.int32)
        ids = np.zeros((len(texts), 32), np.int32)
        ids[:, 0] = [ord(t[0]) for t in texts]
        return ids, lens

    def embed_ids_into(self, store, ids, lens, rows, want_host=False, wait=True):
        vec = np.stack([lens.astype(np.float32), ids[:, 0].astype(np.float32)], axis=1)
        store._collection.put_rows(vec, rows)
        self.pending = not wait

    def wait(self):
        self.pending = False


def test_ingest_chunks_pipeline(monkeypatch):
    """services.indexer.ingest_chunks = _build_payloads + upsert_embeddings in one pass (indexer.py:94-114): both progress
    protocols, the payload mapping, md5 ids and replace-by-primary-key are those of the two reference loops."""
    monkeypatch.setattr(settings, "mi355x_ingest_batch", 64, raising=False)
    root = Path("/w/demo")
    chunks = [Chunk(chr(97 + i % 26) * (i + 1), root / "src" / f"f{i}.py", "python", 10 * i + 1, 10 * i + 9, None) for i in range(130)]
    store = make_store(dim=2)
    store.connect()
    emb, e_seen, u_seen = TokenizingEmbedding(), [], []
    n = ingest_chunks("demo", root, chunks, emb, store, embed_progress=lambda a, b: e_seen.append((a, b)),
                      upsert_progress=lambda a, b: u_seen.append((a, b)))
    assert n == 130 and e_seen == u_seen == [(0, 130), (64, 130), (128, 130), (130, 130)] and emb.tokenized == [64, 64, 2]
    assert emb.pending is False  # batches are enqueued without waiting, and waited for before the last report
    assert len(store) == 130 and store._ids[3] == make_chunk_id("demo", root / "src" / "f3.py", 31, 39)
    assert store._metadata[3] == {"repo": "demo", "path": "src/f3.py", "language": "python", "start_line": 31, "end_line": 39, "symbol": None}
    assert store._texts[3] == "dddd" and store._paths[3] == "src/f3.py" and store._languages[3] == "python"
    assert np.array_equal(store._collection.X[3], [4.0, float(ord("d"))])
    # the same repository again, one chunk edited: nothing is appended, the edited row is replaced (idempotent re-ingest)
    chunks[5] = Chunk("Q" * 7, chunks[5].path, "python", chunks[5].start_line, chunks[5].end_line, None)
    assert ingest_chunks("demo", root, chunks, TokenizingEmbedding(), store) == 130
    assert len(store) == 130 and store._texts[5] == "Q" * 7 and np.array_equal(store._collection.X[5], [7.0, float(ord("Q"))])
    # a primary key that occurs twice keeps its last chunk
    dup = [chunks[0], Chunk("ZZ", chunks[0].path, "python", chunks[0].start_line, chunks[0].end_line, None)]
    seen = []
    assert ingest_chunks("demo", root, dup, TokenizingEmbedding(), store, embed_progress=lambda a, b: seen.append((a, b))) == 1
    assert seen == [(0, 2), (2, 2)] and store._texts[0] == "ZZ" and len(store) == 130
    # empty input: (0, 0) only, nothing is called
    emb, seen = TokenizingEmbedding(), []
    assert ingest_chunks("demo", root, [], emb, store, embed_progress=lambda a, b: seen.append((a, b))) == 0
    assert seen == [(0, 0)] and emb.tokenized == []
    # a failure on the tokenizer thread surfaces in the caller
    with pytest.raises(RuntimeError, match="tokenizer broke"):
        ingest_chunks("demo", root, chunks, TokenizingEmbedding(fail_at=1), store)
