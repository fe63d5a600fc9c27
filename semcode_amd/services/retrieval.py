"""The retrieval half of SemanticSearchPipeline, restated for the MI355X backend.

Reference: src/semcode/rag/pipeline.py:93-175 (`_retrieve_documents`, `_hit_to_document`, `_embed_query`) -- the one
consumer of `MilvusVectorStore.search`'s result shape.  The reference's pipeline runs unchanged
on the two seams; this module exists so that the same behaviour -- lazy connect, top_k from `rag_max_context_sources`, what
counts as "no results", the document mapping and the score attribute order -- can be used and tested without the LLM half of
that class (answer synthesis, prompts: out of scope), and so that MANY questions can be answered by ONE encoder batch and ONE
batched search (`retrieve_batch`), which is how this backend is meant to be driven.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ..settings import resolve as _resolve_settings

log = logging.getLogger(__name__)

_SCORE_ATTRS = ("score", "distance", "similarity")  # pipeline.py:149-155: the first attribute the hit HAS decides


def embed_query(embedding_client: Any, question: str) -> List[float]:
    """pipeline.py:171-175: embed_query when the client has one, else the first row of embed_documents([question])."""
    if hasattr(embedding_client, "embed_query"):
        return embedding_client.embed_query(question)
    return embedding_client.embed_documents([question])[0]


def hit_to_document(hit: Any) -> Optional[Dict[str, Any]]:
    """pipeline.py:131-169: one search hit -> {repo, path, language, snippet, score, metadata}; None for a hit without entity."""
    entity = getattr(hit, "entity", None)
    fetch = getattr(entity, "get", None)
    if entity is None or fetch is None:
        return None
    try:
        repo, path, language = fetch("repo"), fetch("path"), fetch("language")
        snippet = fetch("text")
        metadata = fetch("metadata") or {}
    except Exception:  # an entity with another schema: an empty document, as the reference returns
        repo = path = language = None
        snippet, metadata = "", {}
    score = 0.0
    for attr in _SCORE_ATTRS:
        if hasattr(hit, attr):
            try:
                score = float(getattr(hit, attr))
            except Exception:
                score = 0.0
            break
    return {"repo": repo, "path": path, "language": language, "snippet": snippet or "", "score": score, "metadata": metadata}


def _top_k() -> int:
    return max(1, int(getattr(_resolve_settings(), "rag_max_context_sources", 5)))


class Retriever:
    """question -> documents, with the reference's error protocol: every failure yields [] and is kept in `last_error`
    (`_last_retrieval_error` there); a successful retrieval clears it."""

    def __init__(self, embedding_client: Any, vector_store: Any) -> None:
        self.embedding_client = embedding_client
        self.vector_store = vector_store
        self.last_error: Optional[BaseException] = None
        self._connected = False

    def _ensure_connected(self) -> bool:
        if self._connected:
            return True
        try:
            self.vector_store.connect()
        except Exception as exc:
            log.error("milvus_connection_failed error=%s", exc)
            self.last_error = exc
            return False
        self._connected = True
        return True

    def retrieve(self, question: str) -> List[Dict[str, Any]]:
        """pipeline.py:93-129, one question."""
        if not self._ensure_connected():
            return []
        vector = embed_query(self.embedding_client, question)
        try:
            results = self.vector_store.search(vector, top_k=_top_k())
        except Exception as exc:
            log.error("milvus_search_failed error=%s", exc)
            self.last_error = exc
            return []
        if not results:
            self.last_error = ValueError("no_results")
            return []
        try:
            hits = next(iter(results))
        except StopIteration:
            self.last_error = ValueError("no_results")
            return []
        except TypeError:  # not iterable: taken as the hits themselves, as the reference does
            hits = results
        documents = [doc for doc in (hit_to_document(hit) for hit in hits) if doc]
        self.last_error = None
        return documents

    def retrieve_batch(self, questions: Sequence[str]) -> List[List[Dict[str, Any]]]:
        """Many questions at once: one encoder batch (`embed_documents_array`) and one batched search (`search_batch`) when the
        seams offer them, else `retrieve` per question.  Per question the result is what `retrieve` returns for it (a store
        that returns no hit for a question yields [] for that question)."""
        questions = list(questions)
        if not questions:
            return []
        fast = hasattr(self.embedding_client, "embed_documents_array") and hasattr(self.vector_store, "search_batch") and hasattr(self.vector_store, "hits_for")
        if not fast:
            return [self.retrieve(q) for q in questions]
        if not self._ensure_connected():
            return [[] for _ in questions]
        try:
            vectors = np.asarray(self.embedding_client.embed_documents_array(questions), dtype=np.float32)
            dist, rows = self.vector_store.search_batch(vectors, top_k=_top_k())
            results = self.vector_store.hits_for(dist, rows)
        except Exception as exc:
            log.error("milvus_search_failed error=%s", exc)
            self.last_error = exc
            return [[] for _ in questions]
        out = [[doc for doc in (hit_to_document(hit) for hit in hits) if doc] for hits in results]
        # as retrieve() (pipeline.py:112-122): "no_results" only when the store's result container itself is falsy; a
        # non-empty container whose hit lists hold nothing usable clears the error
        self.last_error = None if results else ValueError("no_results")
        return out
