"""Latency of the query path, one question at a time (pipeline.py:93-175): embed_query -> MilvusVectorStore.search -> hits.

BERT-base shaped encoder with random-init weights and a synthetic vocabulary (nothing real exists offline); the collection is
filled on device with synthetic rows and given scalar columns on the host, sizes as named."""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, ".")
import numpy as np

from scripts.bench_ingest import make_vocab
from semcode_amd.embeddings.providers import MI355XEmbeddings
from semcode_amd.storage import MilvusVectorStore


def timed(fn, reps=30):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t.sort()
    return 1e3 * t[len(t) // 2], 1e3 * t[int(len(t) * 0.9)]


def main():
    tmp = Path(tempfile.mkdtemp(dir="gpurun_out"))
    plain = make_vocab(tmp / "vocab.txt")
    emb = MI355XEmbeddings(vocab=tmp / "vocab.txt", max_tokens=256, allow_synthetic=True)  # random-init benchmark weights
    rng = np.random.default_rng(0)
    for words in (12, 50, 200):
        text = " ".join(plain[i] for i in rng.choice(len(plain), size=words))
        ids, lens = emb.tokenize([text])
        med, p90 = timed(lambda: emb.embed_query(text))
        print(f"embed_query, {int(lens[0]):3d} tokens (bucket {ids.shape[1]:3d}): median {med:6.2f} ms  p90 {p90:6.2f} ms", flush=True)
    q = emb.embed_query("def parse arguments and return the configuration object")
    for rows, kind in ((100_000, "FLAT"), (1_000_000, "FLAT"), (1_000_000, "IVF_FLAT"), (10_000_000, "FLAT"), (10_000_000, "IVF_FLAT")):
        store = MilvusVectorStore(dim=768, metric="IP", index_type=kind)  # reference defaults: IP, nlist 128, nprobe 16
        store.connect()
        store._collection.fill_synthetic(rows, seed=0)
        store._ids = [f"id{i}" for i in range(rows)]
        store._texts = store._repos = store._paths = store._languages = ["x"] * rows
        store._metadata = [{}] * rows
        if kind == "IVF_FLAT":
            t0 = time.perf_counter()
            store.build_index(niter=6)
            extra = f"  (build {time.perf_counter() - t0:.1f} s)"
        else:
            extra = ""
        med, p90 = timed(lambda: next(iter(store.search(q, top_k=5))))
        path = store._collection.last_search_stats()["path"]
        print(f"search top-5, {rows:>10,d} x 768 {kind:8s} [{path}]: median {med:6.2f} ms  p90 {p90:6.2f} ms{extra}", flush=True)
        store.close()
    emb.close()


if __name__ == "__main__":
    main()
