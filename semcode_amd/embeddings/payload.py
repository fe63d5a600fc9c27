"""EmbeddingPayload -- the record passed from the embed stage to the store stage.

Reference: src/semcode/embeddings/providers.py:21-28 (same four fields, same order).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List


@dataclass
class EmbeddingPayload:
    """Embedding representation the storage layer expects."""

    id: str
    text: str
    vector: List[float]
    metadata: dict
