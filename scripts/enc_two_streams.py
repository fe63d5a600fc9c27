"""Experiment: does the encoder gain from two half batches on two streams (tails and launch gaps of one filling the other)?

    python scripts/enc_two_streams.py

One process, one device: (a) one encoder, 256 chunks per forward; (b) two encoders on two runtimes (two streams), 128 chunks per
forward each, driven by two threads.  Prints chunks/s of both."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import bert_oracle as bo  # noqa: E402  (config only)
from semcode_amd import _native  # noqa: E402

cfg = dict(bo.BERT_BASE)
rng = np.random.default_rng(0)
S, steps = 256, 20


def make(batch):
    rt = _native.Runtime(0)
    enc = _native.Encoder(rt, cfg, weights=None, synth_seed=0)
    ids = rng.integers(1000, 30000, size=(batch, S)).astype(np.int32)
    lens = np.full(batch, S, np.int32)
    for _ in range(3):
        enc.embed_ids(ids, lens)
    return rt, enc, ids, lens


def run(enc, ids, lens, n):
    for _ in range(n):
        enc.embed_ids(ids, lens)


rt, enc, ids, lens = make(256)
t = time.time(); run(enc, ids, lens, steps); dt = time.time() - t
print(f"one stream, 256 chunks/forward: {256 * steps / dt:9.0f} chunks/s ({1e3 * dt / steps:.2f} ms/forward, host copies included)")
for batch in (128, 256):
    pair = [make(batch) for _ in range(2)]
    ths = [threading.Thread(target=run, args=(p[1], p[2], p[3], steps)) for p in pair]
    t = time.time()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.time() - t
    print(f"two streams, {batch} chunks/forward each: {2 * batch * steps / dt:9.0f} chunks/s")
    for p in pair:
        p[1].close(); p[0].close()
enc.close(); rt.close()
