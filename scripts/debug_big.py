import sys, time
import numpy as np
sys.path.insert(0, ".")
from semcode_amd import _native
from oracle import sc_oracle as orc

def log(*a):
    print(*a, flush=True)

rt = _native.Runtime(0)
log(rt.device_info())
for N in (200_000, 1_000_000, 4_000_000, 10_000_000):
    ix = _native.Index(rt, 768, metric="L2")
    t = time.time(); ix.fill_synthetic(N, seed=0); rt.synchronize(); log(N, "fill", time.time() - t)
    Q = orc.synth(16, 768, seed=1)
    for rep in range(3):
        t = time.time(); d, r = ix.search(Q, k=10); log(N, "search16", time.time() - t)
    t = time.time(); d, r = ix.search(Q[:1], k=10); log(N, "search1", time.time() - t)
    log(r[0], d[0])
    ix.close()
