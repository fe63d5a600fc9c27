"""GPU: the seams under the reference's threading model.

semcode's API server calls the module-level singletons from a thread pool (reference src/semcode/api/main.py:24-27,160,202):
searches, an ingest job's upserts and embeds may overlap.  Contract: every sc_* call is thread-safe,
calls on one handle are serialised by that handle's lock, ctypes releases the GIL.  Here four threads hammer one store and one
embedding client for a few seconds; afterwards the collection must be exactly what the same upserts produce sequentially."""
import threading
import time

import numpy as np
import pytest

from semcode_amd.embeddings.providers import MI355XEmbeddings
from semcode_amd.storage import MilvusVectorStore

pytestmark = pytest.mark.gpu

SMALL = dict(vocab=2000, hidden=128, layers=2, heads=2, ffn=256, max_pos=128)


def test_concurrent_search_upsert_embed():
    emb = MI355XEmbeddings(cfg=SMALL, synth_seed=7, allow_synthetic=True)
    store = MilvusVectorStore(dim=128, metric="COSINE", index_type="FLAT")
    store.connect()
    rng = np.random.default_rng(0)
    base = rng.standard_normal((4096, 128)).astype(np.float32)
    meta = lambda i: {"repo": "demo", "path": f"src/f{i}.py", "language": "python", "start_line": 1, "end_line": 2, "symbol": None}
    store.upsert_arrays([f"id{i}" for i in range(4096)], base, [f"t{i}" for i in range(4096)], [meta(i) for i in range(4096)])
    batches = []  # what the writer will apply, in order: new keys and replacements of old ones
    for b in range(40):
        keys = [f"id{(b * 37 + j * 11) % 6000}" for j in range(64)]
        keys = list(dict.fromkeys(keys))
        batches.append((keys, rng.standard_normal((len(keys), 128)).astype(np.float32)))
    errors, stop = [], threading.Event()
    counts = {"search": 0, "embed": 0}

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as exc:  # noqa: BLE001 - reported below
                errors.append(exc)
                stop.set()
        return run

    def searcher(seed):
        r = np.random.default_rng(seed)
        while not stop.is_set():
            q = r.standard_normal(128).astype(np.float32)
            hits = next(iter(store.search(q.tolist(), top_k=5)))
            assert len(hits) == 5 and hits[0].distance >= hits[-1].distance and hits[0].entity.get("repo") == "demo"
            d, rows = store.search_batch(r.standard_normal((40, 128)).astype(np.float32), top_k=3)
            assert rows.min() >= 0 and rows.max() < 6000 and np.isfinite(d).all()
            counts["search"] += 1

    def embedder():
        texts = [f"def f{i}(x): return x + {i}" for i in range(48)]
        want = emb.embed_documents_array(texts)
        while not stop.is_set():
            assert np.array_equal(emb.embed_documents_array(texts), want)  # no cross-talk through the shared workspace / stream
            counts["embed"] += 1

    def writer():
        for keys, vec in batches:
            store.upsert_arrays(keys, vec, [f"new {k}" for k in keys], [meta(int(k[2:])) for k in keys])
            time.sleep(0.01)
        stop.set()

    threads = [threading.Thread(target=guard(f)) for f in (lambda: searcher(1), lambda: searcher(2), embedder, writer)]
    t0 = time.time()
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a thread is stuck (lock ordering?)"
    assert not errors, errors
    assert counts["search"] > 5 and counts["embed"] > 5, (counts, time.time() - t0)
    # same upserts applied sequentially to a fresh store: identical rows, columns and search results
    ref = MilvusVectorStore(dim=128, metric="COSINE", index_type="FLAT")
    ref.connect()
    ref.upsert_arrays([f"id{i}" for i in range(4096)], base, [f"t{i}" for i in range(4096)], [meta(i) for i in range(4096)])
    for keys, vec in batches:
        ref.upsert_arrays(keys, vec, [f"new {k}" for k in keys], [meta(int(k[2:])) for k in keys])
    assert len(store) == len(ref) and store._ids == ref._ids and store._texts == ref._texts
    assert np.array_equal(store._collection.get_rows(0, len(ref)), ref._collection.get_rows(0, len(ref)))
    q = rng.standard_normal((64, 128)).astype(np.float32)
    (da, ra), (db, rb) = store.search_batch(q, top_k=10), ref.search_batch(q, top_k=10)
    assert np.array_equal(ra, rb) and np.array_equal(da, db)
    for s in (store, ref):
        s.close()
    emb.close()


def test_two_runtimes_in_one_process_interleave_searches():
    """Several runtimes may live in one process (a service that owns a sharded collection: one runtime per device).  The launch
    sites used to raise a kernel's dynamic-LDS limit behind process-wide flags; they are keyed by device now (sc_device_once).  This
    box has one GPU, so the test runs two runtimes -- two streams -- on device 0 from two threads, every search path (exact, batched
    int8 / bf16, IVF_FLAT probes), and checks each result against the CPU oracle."""
    from oracle import sc_oracle as orc
    from semcode_amd import _native

    rts = [_native.Runtime(device=0), _native.Runtime(device=0)]
    X = [orc.synth(30_000, 256, seed=61), orc.synth(20_000, 256, seed=62)]
    Q = orc.synth(48, 256, seed=63)
    want = [orc.search(x, Q, 10, "L2") for x in X]
    idx = []
    for rt, x in zip(rts, X):
        ix = _native.Index(rt, 256, metric="L2")
        ix.add(x)
        idx.append(ix)
    errors = []

    def worker(i):
        try:
            ix = idx[i]
            od, orow = want[i]
            for rep in range(6):
                for mode, nq in (("exact", 9), ("batched", 48), ("exact", 16), ("batched", 40)):
                    ix.set_search_mode(mode)
                    ix.set_coarse_stage(8 if rep % 2 else 16)
                    d, r = ix.search(Q[:nq], k=10)
                    assert np.array_equal(r, orow[:nq]) and np.array_equal(d.view(np.uint32), od[:nq].view(np.uint32)), (i, rep, mode)
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for ix in idx:
        ix.close()
    for rt in rts:
        rt.close()
    assert not errors, errors[0]
