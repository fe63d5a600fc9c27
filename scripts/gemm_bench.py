import sys
sys.path.insert(0, ".")
from semcode_amd import _native
rt = _native.Runtime(0)
shapes = [(65536, 2304, 768, 0, "qkv"), (65536, 768, 768, 2, "out"), (65536, 3072, 768, 1, "ffn1"), (65536, 768, 3072, 2, "ffn2"),
          (8192, 8192, 8192, 0, "8k^3"), (4096, 4096, 4096, 0, "4k^3")]
variants = [int(v) for v in sys.argv[1:]] or [0, 128]
for M, N, K, epi, name in shapes:
    row = []
    for v in variants:
        e = epi if (v in (0, 128) or v >= 1000) else 0  # ablations: 1 no DMA, 2 no MFMA, 4 no epilogue, 5 = 1+4, 21 = MFMA only
        ms = _native.diag_gemm_bench(rt, M, N, K, epi=e, iters=10, variant=v)
        row.append(f"v{v}: {ms*1e3:8.1f}us {2.0*M*N*K/ms/1e9:7.1f}TF")
    print(f"{name:6s} M={M} N={N} K={K} epi={epi} | " + " | ".join(row), flush=True)
