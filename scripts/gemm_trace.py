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

# ---- where does the spread of the main-loop time come from?  by XCD, by dispatch round on a CU, by tile column
M, N, K, epi, name = 65536, 2304, 768, 0, "qkv"
T = _native.diag_gemm_trace(rt, M, N, K, epi=epi, launches=3).astype(np.int64)[1]
hw, xcc = T[:, 0], T[:, 1] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
ml = (T[:, 3] - T[:, 2]) / 100.0
ent = T[:, 2] - T[:, 2].min()
print(f"{name}: main loop us by XCD:", " ".join(f"{x}:{ml[xcc == x].mean():.2f}" for x in range(8)))
rounds = np.zeros(len(T), int)
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    rounds[idx[np.argsort(ent[idx])]] = np.arange(len(idx))
print("   by dispatch round on its CU:", " ".join(f"{r}:{ml[rounds == r].mean():.2f}" for r in range(rounds.max() + 1)))
tiles_n = N // 256
b = np.arange(len(T))  # logical tile of a workgroup: XCD remap + column-major inside groups of 8 row panels (gemm_bf16.hip tile_coords256)
q, r = len(T) // 8, len(T) % 8
x = b & 7
tile = np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + (b >> 3)
col = (tile % (8 * tiles_n)) // 8
print("   by tile column inside its group of 8 row panels:", " ".join(f"{c}:{ml[col == c].mean():.2f}" for c in range(tiles_n)))
tot = np.array([ml[cu == c].sum() for c in np.unique(cu)])
print(f"   sum of main loops per CU: mean {tot.mean():.1f} us  min {tot.min():.1f}  max {tot.max():.1f}  (sd of one tile {ml.std():.2f})")
rt.close()
