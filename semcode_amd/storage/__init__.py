"""Vector storage seam (drop-in for semcode.storage.MilvusVectorStore)."""
from .milvus_store import Hit, Hits, MilvusVectorStore, SearchResult

__all__ = ["MilvusVectorStore", "SearchResult", "Hits", "Hit"]
