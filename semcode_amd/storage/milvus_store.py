"""MilvusVectorStore drop-in backed by the HBM-resident index of libsemcode_hip.

Reference: src/semcode/storage/milvus_store.py:29-148.  Same constructor, attributes, methods,
progress protocol and error behaviour, so IndexerService (src/semcode/services/indexer.py:54-63,
108) and SemanticSearchPipeline (src/semcode/rag/pipeline.py:93-169) run on it unchanged:

    MilvusVectorStore(collection_name="semcode_chunks", dim=None)
    .connect()                                  -> opens the device runtime + index (was: gRPC connect,
                                                   create collection + IVF_FLAT/IP index, load)
    .upsert_embeddings(payloads, progress=None) -> replace-by-primary-key, batches of
                                                   settings.milvus_upsert_batch_size
    .search(vector, top_k=10)                   -> iterable of Hits; hit.entity.get(field), hit.distance,
                                                   hit.score, hit.id  (pymilvus SearchResult shape)

What lives where: vectors and their norms in HBM (sc_index); the scalar columns of the reference
schema (id, repo, path, language, text, metadata; milvus_store.py:59-74) and the md5 -> row map in
this object.  Additions that do not break the reference surface: search_batch(), metric/index options.
"""
from __future__ import annotations

import json
import logging
import shutil
import threading
from pathlib import Path
from typing import Any, Callable, Iterator, List, Optional, Sequence

import numpy as np

from ..embeddings.payload import EmbeddingPayload
from ..settings import resolve as _resolve_settings

log = logging.getLogger(__name__)

OUTPUT_FIELDS = ("repo", "path", "language", "text", "metadata")  # milvus_store.py:146


class _Entity:
    """pymilvus Hit.entity look-alike: .get(name) over the stored scalar columns."""

    __slots__ = ("_fields",)

    def __init__(self, fields: dict) -> None:
        self._fields = fields

    def get(self, name: str, default: Any = None) -> Any:
        return self._fields.get(name, default)

    def to_dict(self) -> dict:
        return dict(self._fields)


class Hit:
    """One search hit: .id (primary key), .distance == .score (metric value), .entity.get(field)."""

    __slots__ = ("id", "distance", "entity", "row")

    def __init__(self, pk: str, distance: float, fields: dict, row: int) -> None:
        self.id = pk
        self.distance = float(distance)
        self.entity = _Entity(fields)
        self.row = row

    @property
    def score(self) -> float:
        return self.distance

    def __repr__(self) -> str:  # pragma: no cover
        return f"Hit(id={self.id!r}, distance={self.distance:.6g})"


class Hits(list):
    """Per-query hit list, best first (pymilvus Hits)."""

    @property
    def ids(self) -> list:
        return [h.id for h in self]

    @property
    def distances(self) -> list:
        return [h.distance for h in self]


class SearchResult(list):
    """List of Hits, one per query; `next(iter(result))` is what pipeline.py:117-118 does."""


class MilvusVectorStore:
    """Thin wrapper with the reference's surface, storing vectors on the MI355X."""

    def __init__(self, collection_name: str = "semcode_chunks", dim: Optional[int] = None, *, metric: Optional[str] = None,
                 index_type: Optional[str] = None, nlist: Optional[int] = None, nprobe: Optional[int] = None,
                 device: Optional[int] = None, runtime: Any = None, index_factory: Optional[Callable[..., Any]] = None) -> None:
        settings = _resolve_settings()
        self.collection_name = collection_name
        self.dim = dim or settings.embedding_dimension
        self._collection: Any = None  # the device index once connected (name kept from the reference)
        self.metric = (metric or getattr(settings, "mi355x_metric", "IP")).upper()
        self.index_type = (index_type or getattr(settings, "mi355x_index_type", "IVF_FLAT")).upper()
        self.nlist = int(nlist or getattr(settings, "mi355x_nlist", 128))
        self.nprobe = int(nprobe or getattr(settings, "mi355x_nprobe", 16))
        self._device = int(device if device is not None else getattr(settings, "mi355x_device", 0))
        self._runtime = runtime
        self._owns_runtime = runtime is None
        self._index_factory = index_factory
        self._lock = threading.RLock()
        # scalar columns, indexed by row (milvus_store.py:59-74)
        self._ids: List[str] = []
        self._texts: List[str] = []
        self._metadata: List[dict] = []
        self._repos: List[str] = []
        self._paths: List[str] = []
        self._languages: List[str] = []
        self._row_of: dict[str, int] = {}
        self._needs_train = False  # IVF_FLAT lists are (re)built lazily before a search, like Milvus' background index build

    # ------------------------------------------------------------------ lifecycle
    def connect(self) -> None:
        """Open the device and create the (empty) collection.  May raise; callers catch Exception."""
        settings = _resolve_settings()
        log.info("connecting_mi355x device=%s (milvus_uri %s is not used)", self._device, getattr(settings, "milvus_uri", None))
        with self._lock:
            if self._collection is not None:
                return
            self._collection = self._ensure_collection()
            store_path = getattr(settings, "mi355x_store_path", None)
            if store_path and (Path(store_path) / "manifest.json").exists() and not self._ids:
                self.load(store_path)  # utility.has_collection(...) -> Collection(name).load() of the reference (milvus_store.py:51-54)

    def _ensure_collection(self) -> Any:
        if self._index_factory is not None:
            return self._index_factory(dim=self.dim, metric=self.metric, kind=self.index_type, nlist=self.nlist)
        from .. import _native  # raises loudly if libsemcode_hip.so is missing

        if self._runtime is None:
            self._runtime = _native.shared_runtime(self._device)  # shared with the embedding client (device-to-device upserts)
            self._owns_runtime = False
        log.info("creating_collection %s dim=%d metric=%s index=%s", self.collection_name, self.dim, self.metric, self.index_type)
        return _native.Index(self._runtime, self.dim, metric=self.metric, kind=self.index_type, nlist=self.nlist)

    def close(self) -> None:
        with self._lock:
            if self._collection is not None and hasattr(self._collection, "close"):
                self._collection.close()
            self._collection = None
            if self._owns_runtime and self._runtime is not None:
                self._runtime.close()
                self._runtime = None

    def __len__(self) -> int:
        return len(self._ids)

    # ------------------------------------------------------------------ upsert
    def upsert_embeddings(self, payloads: Sequence[EmbeddingPayload], progress: Optional[Callable[[int, int], None]] = None) -> None:
        """Insert or update embeddings (replace by primary key), reference milvus_store.py:87-133."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")

        payload_list: List[EmbeddingPayload] = list(payloads)
        total = len(payload_list)
        log.info("upserting_embeddings count=%d", total)
        if progress:
            progress(0, total)
        if total == 0:
            return

        settings = _resolve_settings()
        batch_size = max(1, getattr(settings, "milvus_upsert_batch_size", 128))
        inserted = 0
        for start in range(0, total, batch_size):
            batch = payload_list[start:start + batch_size]
            with self._lock:
                self._upsert_batch(batch)
            inserted += len(batch)
            if progress:
                progress(inserted, total)
        self._needs_train = True

    def _upsert_batch(self, batch: Sequence[EmbeddingPayload]) -> None:
        vectors = np.asarray([p.vector for p in batch], dtype=np.float32)
        if vectors.ndim != 2 or vectors.shape[1] != self.dim:
            raise ValueError(f"embedding dimension mismatch: collection dim={self.dim}, got array of shape {vectors.shape}")
        # a primary key repeated inside one batch: the last occurrence wins (upsert semantics)
        last = {p.id: i for i, p in enumerate(batch)}
        new_idx, new_rows, old_idx, old_rows = [], [], [], []
        next_row = len(self._ids)
        for pk, i in last.items():
            row = self._row_of.get(pk)
            if row is None:
                new_idx.append(i)
                new_rows.append(next_row)
                next_row += 1
            else:
                old_idx.append(i)
                old_rows.append(row)
        order = np.argsort(new_idx) if new_idx else []
        new_idx = [new_idx[j] for j in order]
        if new_idx:
            self._collection.add(vectors[new_idx])
        if old_idx:
            self._collection.overwrite(vectors[old_idx], np.asarray(old_rows, dtype=np.int64))
        for i in new_idx:
            p = batch[i]
            self._row_of[p.id] = len(self._ids)
            self._ids.append(p.id)
            self._repos.append(p.metadata.get("repo", ""))
            self._paths.append(p.metadata.get("path", ""))
            self._languages.append(p.metadata.get("language", ""))
            self._texts.append(p.text)
            self._metadata.append(p.metadata)
        for i, row in zip(old_idx, old_rows):
            p = batch[i]
            self._repos[row] = p.metadata.get("repo", "")
            self._paths[row] = p.metadata.get("path", "")
            self._languages[row] = p.metadata.get("language", "")
            self._texts[row] = p.text
            self._metadata[row] = p.metadata

    # ------------------------------------------------------------------ array fast paths
    def plan_rows(self, ids: Sequence[str]) -> np.ndarray:
        """Row numbers an upsert of these primary keys writes: the existing row of a known key, else the next free rows in
        order of appearance.  Keys must be distinct (deduplicate first: the last occurrence wins in an upsert)."""
        if len(set(ids)) != len(ids):
            raise ValueError("plan_rows: primary keys must be distinct")
        rows = np.empty(len(ids), dtype=np.int64)
        next_row = len(self._ids)
        for i, pk in enumerate(ids):
            row = self._row_of.get(pk)
            if row is None:
                row = next_row
                next_row += 1
            rows[i] = row
        return rows

    def commit_rows(self, ids: Sequence[str], rows: np.ndarray, texts: Sequence[str], metadatas: Sequence[dict]) -> None:
        """Scalar columns + primary-key map for rows whose vectors have just been written (same mapping as _upsert_batch)."""
        for pk, row, text, meta in zip(ids, rows.tolist(), texts, metadatas):
            cols = (meta.get("repo", ""), meta.get("path", ""), meta.get("language", ""))
            if row == len(self._ids):
                self._row_of[pk] = row
                self._ids.append(pk)
                self._repos.append(cols[0])
                self._paths.append(cols[1])
                self._languages.append(cols[2])
                self._texts.append(text)
                self._metadata.append(meta)
            else:
                self._repos[row], self._paths[row], self._languages[row] = cols
                self._texts[row] = text
                self._metadata[row] = meta
        self._needs_train = True

    def upsert_arrays(self, ids: Sequence[str], vectors: Any, texts: Sequence[str], metadatas: Sequence[dict],
                      progress: Optional[Callable[[int, int], None]] = None) -> None:
        """upsert_embeddings without EmbeddingPayload / list[float] boxing: vectors is one [n, dim] float array.  Same
        progress protocol and batch size (settings.milvus_upsert_batch_size), same replace-by-primary-key result."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        vec = np.asarray(vectors, dtype=np.float32)
        total = len(ids)
        if vec.ndim != 2 or vec.shape != (total, self.dim):
            raise ValueError(f"embedding dimension mismatch: collection dim={self.dim}, expected [{total}, {self.dim}], got {vec.shape}")
        if not (len(texts) == len(metadatas) == total):
            raise ValueError("ids, texts and metadatas must have one entry per vector")
        if progress:
            progress(0, total)
        if total == 0:
            return
        batch_size = max(1, getattr(_resolve_settings(), "milvus_upsert_batch_size", 128))
        done = 0
        for start in range(0, total, batch_size):
            stop = min(total, start + batch_size)
            last = {pk: i for i, pk in enumerate(ids[start:stop], start)}  # repeated key inside a batch: last wins
            keep = sorted(last.values())
            b_ids = [ids[i] for i in keep]
            with self._lock:
                rows = self.plan_rows(b_ids)
                self._collection.put_rows(vec[keep], rows)
                self.commit_rows(b_ids, rows, [texts[i] for i in keep], [metadatas[i] for i in keep])
            done = stop
            if progress:
                progress(done, total)

    def upsert_encoded(self, ids: Sequence[str], token_ids: np.ndarray, lens: np.ndarray, texts: Sequence[str], metadatas: Sequence[dict],
                       embedding_client: Any, wait: bool = True) -> None:
        """One batch, embed + upsert fused: the encoder output goes from its device buffer straight into the index rows.
        wait=False: the device work is only enqueued (embedding_client.wait() completes it); searches issued afterwards are
        ordered behind it on the device stream either way."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        with self._lock:
            rows = self.plan_rows(ids)
            embedding_client.embed_ids_into(self, token_ids, lens, rows, wait=wait)
            self.commit_rows(ids, rows, texts, metadatas)

    # ------------------------------------------------------------------ search
    def search(self, vector: "list[float]", top_k: int = 10) -> SearchResult:
        """Run a raw vector search (one query), reference milvus_store.py:135-148."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        dist, rows = self.search_batch(np.asarray([vector], dtype=np.float32), top_k)
        return SearchResult([self._hits(dist[0], rows[0])])

    def search_batch(self, queries: Any, top_k: int = 10) -> "tuple[np.ndarray, np.ndarray]":
        """Batched search: queries [Q, dim] -> (dist [Q, k] f32, rows [Q, k] i64; -1 = no hit), best first."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        q = np.asarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"query dimension mismatch: collection dim={self.dim}, got shape {q.shape}")
        with self._lock:
            self._maybe_train()
            return self._collection.search(q, k=int(top_k), nprobe=self.nprobe)

    def build_index(self, niter: int = 10) -> None:
        """(Re)build the IVF_FLAT lists now (create_index + load of the reference, milvus_store.py:76-84)."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        with self._lock:
            if self.index_type == "IVF_FLAT" and hasattr(self._collection, "train") and len(self._ids) > 0:
                self._collection.train(niter=niter)
            self._needs_train = False

    def _maybe_train(self) -> None:
        # faiss' rule of thumb: at least 39 points per centroid, otherwise the exhaustive scan is used (exact results)
        if self._needs_train and self.index_type == "IVF_FLAT" and hasattr(self._collection, "train") and len(self._ids) >= 39 * self.nlist:
            self._collection.train(niter=10)
        self._needs_train = False

    def hits_for(self, dist: np.ndarray, rows: np.ndarray) -> SearchResult:
        """Materialise pymilvus-shaped results for a search_batch() output."""
        return SearchResult([self._hits(d, r) for d, r in zip(dist, rows)])

    def _hits(self, dist: np.ndarray, rows: np.ndarray) -> Hits:
        hits = Hits()
        for d, r in zip(dist.tolist(), rows.tolist()):
            if r < 0:
                continue
            fields = {"repo": self._repos[r], "path": self._paths[r], "language": self._languages[r], "text": self._texts[r],
                      "metadata": self._metadata[r]}
            hits.append(Hit(self._ids[r], d, fields, r))
        return hits

    # ------------------------------------------------------------------ persistence (replaces Milvus' volume, docker-compose.yml:13-14)
    def save(self, path: "str | Path") -> None:
        """Write the collection to `path/` : manifest.json, vectors.f32 (row-major [rows, dim]), columns.jsonl
        (id, repo, path, language, text, metadata per row) and, when the IVF_FLAT lists are built, ivf_centroids.f32
        ([nlist, dim]) + ivf_assign.i32 (list of every row).  Written to a temporary directory and renamed."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        path = Path(path)
        tmp = path.with_name(path.name + ".tmp")
        if tmp.exists():
            shutil.rmtree(tmp)
        tmp.mkdir(parents=True)
        with self._lock:
            n = len(self._ids)
            with open(tmp / "vectors.f32", "wb") as f:
                for start in range(0, n, 65536):
                    m = min(65536, n - start)
                    np.ascontiguousarray(self._collection.get_rows(start, m), dtype="<f4").tofile(f)
            with open(tmp / "columns.jsonl", "w", encoding="utf-8") as f:
                for r in range(n):
                    f.write(json.dumps({"id": self._ids[r], "repo": self._repos[r], "path": self._paths[r], "language": self._languages[r],
                                        "text": self._texts[r], "metadata": self._metadata[r]}, ensure_ascii=False) + "\n")
            manifest = {"format": "semcode_amd.collection.v1", "collection_name": self.collection_name, "dim": self.dim, "rows": n,
                        "metric": self.metric, "index_type": self.index_type, "nlist": self.nlist, "nprobe": self.nprobe}
            ivf = self._collection.ivf_info() if (not self._needs_train and hasattr(self._collection, "ivf_info")) else None
            if ivf and ivf.get("nlist", 0) > 0 and hasattr(self._collection, "ivf_assignments"):
                np.ascontiguousarray(ivf["centroids"], dtype="<f4").tofile(tmp / "ivf_centroids.f32")
                np.ascontiguousarray(self._collection.ivf_assignments(), dtype="<i4").tofile(tmp / "ivf_assign.i32")
                manifest["ivf_trained_nlist"] = int(ivf["nlist"])
            (tmp / "manifest.json").write_text(json.dumps(manifest, indent=1))
        if path.exists():
            shutil.rmtree(path)
        tmp.rename(path)

    def load(self, path: "str | Path") -> None:
        """Replace the collection's content with a directory written by save() (connect() first)."""
        if self._collection is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        path = Path(path)
        manifest = json.loads((path / "manifest.json").read_text())
        if manifest.get("format") != "semcode_amd.collection.v1":
            raise ValueError(f"{path}: unknown collection format {manifest.get('format')!r}")
        if manifest["dim"] != self.dim:
            raise ValueError(f"{path}: stored dim {manifest['dim']} != collection dim {self.dim}")
        with self._lock:
            if len(self._ids):
                raise RuntimeError("load() needs an empty collection")
            n = manifest["rows"]
            vec = np.memmap(path / "vectors.f32", dtype="<f4", mode="r", shape=(n, self.dim)) if n else np.zeros((0, self.dim), np.float32)
            for start in range(0, n, 65536):
                self._collection.add(np.asarray(vec[start:start + 65536], dtype=np.float32))
            with open(path / "columns.jsonl", encoding="utf-8") as f:
                for line in f:
                    c = json.loads(line)
                    self._row_of[c["id"]] = len(self._ids)
                    self._ids.append(c["id"])
                    self._repos.append(c["repo"])
                    self._paths.append(c["path"])
                    self._languages.append(c["language"])
                    self._texts.append(c["text"])
                    self._metadata.append(c["metadata"])
            if len(self._ids) != n:
                raise ValueError(f"{path}: columns.jsonl holds {len(self._ids)} rows, manifest says {n}")
            self._needs_train = True
            # the saved lists are reused as they are (no k-means) when they fit this collection's index parameters
            tn = int(manifest.get("ivf_trained_nlist", 0))
            if (tn > 0 and n > 0 and self.index_type == "IVF_FLAT" and manifest.get("metric") == self.metric and hasattr(self._collection, "set_ivf")
                    and (path / "ivf_centroids.f32").exists() and (path / "ivf_assign.i32").exists()):
                cent = np.fromfile(path / "ivf_centroids.f32", dtype="<f4")
                assign = np.fromfile(path / "ivf_assign.i32", dtype="<i4")
                if cent.size == tn * self.dim and assign.size == n:
                    self._collection.set_ivf(cent.reshape(tn, self.dim), assign)
                    self._needs_train = False

    def __iter__(self) -> Iterator:  # pragma: no cover - convenience
        return iter(self._ids)
