"""Embedding seam (drop-in for semcode.embeddings)."""
from .payload import EmbeddingPayload

__all__ = ["EmbeddingPayload"]
