"""Embedding seam (drop-in for semcode.embeddings): same two public names as the reference package
(src/semcode/embeddings/__init__.py)."""
from .payload import EmbeddingPayload
from .providers import EmbeddingProviderFactory

__all__ = ["EmbeddingPayload", "EmbeddingProviderFactory"]
