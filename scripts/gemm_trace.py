"""Per-workgroup timeline of the 256x256-tile GEMM: where does a CU's time go between main loops?"""
import sys
sys.path.insert(0, ".")
import numpy as np
from semcode_amd import _native

rt = _native.Runtime(0)
shapes = [(65536, 2304, 768, 0, "qkv"), (65536, 2304, 768, 1, "qkv+gelu"), (65536, 3072, 768, 0, "ffn1-nogelu"), (65536, 3072, 768, 1, "ffn1"),
          (65536, 768, 768, 2, "out"), (65536, 768, 3072, 2, "ffn2")]
for M, N, K, epi, name in shapes:
    T = _native.diag_gemm_trace(rt, M, N, K, epi=epi, launches=3).astype(np.int64)
    base = T[0, :, 2].min()
    edges = [(T[l, :, 2].min() - base, T[l, :, 5].max() - base) for l in range(3)]
    t = T[1]
    hw, xcc = t[:, 0], t[:, 1] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # cu, sh, se, xcc
    t0 = t[:, 2].min()
    ent, ml, ep, dr = (t[:, i] - t0 for i in (2, 3, 4, 5))
    print(f"{name}: tiles {len(t)}  distinct CUs {len(np.unique(cu))}  span {(dr.max()) / 100:.1f} us;  launches [first entry, last drain] us: "
          + "  ".join(f"[{a / 100:.1f}, {b / 100:.1f}]" for a, b in edges))
    q = lambda x: f"mean {np.mean(x) / 100:6.2f} us  p10 {np.percentile(x, 10) / 100:6.2f}  p90 {np.percentile(x, 90) / 100:6.2f}"
    print(f"   main loop (entry->done)   {q(ml - ent)}")
    print(f"   epilogue issue            {q(ep - ml)}")
    print(f"   store drain               {q(dr - ep)}")
    gaps, busy = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(ent[idx])]
        gaps += list((ent[idx][1:] - dr[idx][:-1]))
        busy.append(len(idx))
    print(f"   gap drained->next entry   {q(np.array(gaps))}   (n={len(gaps)})  tiles per CU min {min(busy)} max {max(busy)}")
    first = np.sort(ent)[:256]
    print(f"   first-round entry spread  {first.max() / 100:.2f} us;  last-round drain spread {(np.sort(dr)[-256:].max() - np.sort(dr)[-256:].min()) / 100:.2f} us")
    sys.stdout.flush()
rt.close()
