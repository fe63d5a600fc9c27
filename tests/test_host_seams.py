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
from semcode_amd.services import build_payloads, make_chunk_id
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

    def get_rows(self, first, n):
        return self.X[first:first + n].copy()

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
    assert [c for c in s._collection.calls] == [("add", 128), ("add", 128), ("add", 44)]
    seen.clear()
    s.upsert_embeddings([], progress=lambda a, b: seen.append((a, b)))
    assert seen == [(0, 0)]


def test_upsert_replaces_by_primary_key():
    s = make_store()
    s.connect()
    s.upsert_embeddings([payload(0, [1, 0, 0, 0]), payload(1, [0, 1, 0, 0])])
    s.upsert_embeddings([payload(1, [0, 0, 1, 0]), payload(2, [0, 0, 0, 1]), payload(2, [0, 0, 0, 2])])
    assert len(s) == 3
    assert ("overwrite", [1]) in s._collection.calls
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
