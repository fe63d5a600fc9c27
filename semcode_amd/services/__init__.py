"""Host-side counterparts of the reference callers on the hot path."""
from .indexer import build_payloads, embedding_batch_size, ingest_chunks, make_chunk_id
from .retrieval import Retriever, embed_query, hit_to_document

__all__ = ["build_payloads", "make_chunk_id", "embedding_batch_size", "ingest_chunks", "Retriever", "embed_query", "hit_to_document"]
