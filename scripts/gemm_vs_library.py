"""Same-box comparison: the product's 256-tile GEMM against the vendor library (torch.matmul -> hipBLASLt/rocBLAS), bf16 in,
f32 accumulate, at the four encoder shapes and two square ones.  The library is measured ONLY as a yardstick for DESIGN.md §4; the
product path never calls it.  Run on the GPU box:  python3 scripts/gemm_vs_library.py
"""
import sys

sys.path.insert(0, ".")
import torch

from semcode_amd import _native

rt = _native.Runtime(0)
shapes = [(65536, 2304, 768, 0, "qkv"), (65536, 768, 768, 2, "out"), (65536, 3072, 768, 1, "ffn1"), (65536, 768, 3072, 2, "ffn2"),
          (8192, 8192, 8192, 0, "8k^3"), (4096, 4096, 4096, 0, "4k^3")]
dev = torch.device("cuda:0")


def lib_ms(M, N, K, with_bias, iters=10):
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    f = (lambda: torch.nn.functional.linear(a, w, b)) if with_bias else (lambda: a @ w.t())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, N, K, epi, name in shapes:
    ours_plain = _native.diag_gemm_bench(rt, M, N, K, epi=0, iters=10, variant=0)
    ours_epi = _native.diag_gemm_bench(rt, M, N, K, epi=epi, iters=10, variant=0)
    lib_plain = lib_ms(M, N, K, False)
    lib_bias = lib_ms(M, N, K, True)
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:5s} M={M} N={N} K={K} | ours bias-only {ours_plain*1e3:7.1f}us {fl/ours_plain:7.1f}TF | ours epi={epi} {ours_epi*1e3:7.1f}us "
          f"{fl/ours_epi:7.1f}TF | library matmul {lib_plain*1e3:7.1f}us {fl/lib_plain:7.1f}TF | library linear+bias {lib_bias*1e3:7.1f}us "
          f"{fl/lib_bias:7.1f}TF", flush=True)
