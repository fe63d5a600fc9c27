"""Host-side counterparts of the reference callers on the hot path."""
from .indexer import build_payloads, embedding_batch_size, ingest_chunks, make_chunk_id

__all__ = ["build_payloads", "make_chunk_id", "embedding_batch_size", "ingest_chunks"]
