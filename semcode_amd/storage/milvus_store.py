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

    RETRAIN_GROWTH = 2.0  # re-run k-means when the collection has this many times the rows its centroids were trained on

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
        # IVF_FLAT lists are built lazily before a search (Milvus' background index build).  Once built they are kept across
        # upserts: the device index assigns upserted rows to the existing centroids at the next search (no k-means, like
        # Collection.upsert into an indexed collection, milvus_store.py:128).  k-means runs again only on build_index() or when
        # the collection has grown to RETRAIN_GROWTH x the row count the centroids were trained on.
        self._needs_train = False
        self._trained_rows = 0

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

    def _upsert_batch(self, batch: Sequence[EmbeddingPayload]) -> None:
        vectors = np.asarray([p.vector for p in batch], dtype=np.float32)
        if vectors.ndim != 2 or vectors.shape[1] != self.dim:
            raise ValueError(f"embedding dimension mismatch: collection dim={self.dim}, got array of shape {vectors.shape}")
        # a primary key repeated inside one batch: the last occurrence wins (upsert semantics)
        keep = sorted({p.id: i for i, p in enumerate(batch)}.values())
        ids = [batch[i].id for i in keep]
        rows = self.plan_rows(ids)
        # one native call per batch: it validates that every row is an existing row or the next free one, so a drift between
        # the device index and the host columns fails loudly; the columns are committed only after it has succeeded
        self._put_rows(vectors[keep], rows)
        self.commit_rows(ids, rows, [batch[i].text for i in keep], [batch[i].metadata for i in keep])

    def _put_rows(self, vectors: np.ndarray, rows: np.ndarray) -> None:
        ix = self._collection
        if hasattr(ix, "put_rows"):
            ix.put_rows(vectors, rows)
        else:  # an index_factory object with only the add / overwrite pair
            new = rows >= len(self._ids)
            if new.any():
                ix.add(vectors[new])
            if (~new).any():
                ix.overwrite(vectors[~new], rows[~new])
        if hasattr(ix, "__len__") and len(ix) != max(len(self._ids), int(rows.max()) + 1):
            raise RuntimeError(f"vector index holds {len(ix)} rows, the collection's columns expect {max(len(self._ids), int(rows.max()) + 1)}")

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
        self._note_growth()

    def _note_growth(self) -> None:
        n = len(self._ids)
        if self._trained_rows == 0 or n >= self.RETRAIN_GROWTH * self._trained_rows:
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
                self._put_rows(vec[keep], rows)
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
            # argument errors surface here, before anything is committed; with wait=False a failure of the enqueued device
            # work is reported by embedding_client.wait() (services.ingest_chunks calls it before it returns) -- the rows of
            # that batch then hold undefined vectors and the caller must upsert them again
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
                self._train(niter)
            self._needs_train = False

    def _train(self, niter: int) -> None:
        if hasattr(self._collection, "release_scratch"):
            self._collection.release_scratch()  # training needs a second copy of the corpus: drop what can be rebuilt first
        self._collection.train(niter=niter)
        self._trained_rows = len(self._ids)

    def _maybe_train(self) -> None:
        # faiss' rule of thumb: at least 39 points per centroid, otherwise the exhaustive scan is used (exact results)
        if self._needs_train and self.index_type == "IVF_FLAT" and hasattr(self._collection, "train") and len(self._ids) >= 39 * self.nlist:
            try:
                self._train(10)
            except Exception as exc:  # e.g. no room for the second corpus copy: serve exact exhaustive results instead of failing every search
                log.warning("ivf_train_failed rows=%d nlist=%d: %s -- answering with the exhaustive scan until build_index() succeeds",
                            len(self._ids), self.nlist, exc)
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
                    self._trained_rows = n

    def __iter__(self) -> Iterator:  # pragma: no cover - convenience
        return iter(self._ids)
