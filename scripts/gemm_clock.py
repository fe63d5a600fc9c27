"""Cycles and clock of the 256-tile main loop, per ablation: SC_GEMM_PP / SC_GEMM_TRACE_DBG select the loop and what it leaves out.

    python scripts/gemm_clock.py            (one process per variant: the switches are read once)
"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    import numpy as np
    from semcode_amd import _native

    rt = _native.Runtime(0)
    for M, N, K, name in [(8192, 8192, 8192, "8k^3"), (65536, 2304, 768, "qkv"), (65536, 768, 3072, "ffn2")]:
        T = _native.diag_gemm_trace(rt, M, N, K, epi=0, launches=4).astype(np.int64)[1:]
        wall = (T[:, :, 3] - T[:, :, 2]) * 10e-9  # seconds (100 MHz stamps)
        cyc = (T[:, :, 7] - T[:, :, 6]).astype(np.float64)
        ok = (wall > 0) & (cyc > 0)
        nk = K // 64
        print(f"  {name:5s} main loop per tile {wall[ok].mean() * 1e6:7.2f} us  {cyc[ok].mean():9.0f} cycles  clock {np.median(cyc[ok] / wall[ok]) / 1e9:5.3f} GHz  "
              f"cycles per K-tile {cyc[ok].mean() / nk:7.1f}  (MFMA issue alone: 2048)", flush=True)
    rt.close()
    sys.exit(0)

for pp, dbg, what in [(4, 4, "ping-pong, full"), (4, 5, "ping-pong, no LDS-DMA in the loop"), (4, 21, "ping-pong, MFMAs + barriers only"), (4, 20, "ping-pong, no fragment reads"),
                      (0, 4, "one barrier per K-tile, full"), (0, 5, "one barrier, no LDS-DMA"), (0, 21, "one barrier, MFMAs only")]:
    env = dict(os.environ, SC_GEMM_PP=str(pp), SC_GEMM_TRACE_DBG=str(dbg))
    print(f"{what} (SC_GEMM_PP={pp}, dbg {dbg})", flush=True)
    subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
