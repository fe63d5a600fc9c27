"""More seeds of tests/test_encoder_fuzz_gpu.py in one process (GPU box): python tests/encoder_fuzz_more.py [first] [last]."""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import importlib.util, numpy as np
from semcode_amd import _native
spec = importlib.util.spec_from_file_location("f", "tests/test_encoder_fuzz_gpu.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rt = _native.Runtime(device=0)
fn = m.test_random_encoder_shapes_match_restatement
fn = getattr(fn, "__wrapped__", fn)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 124):
    try:
        fn(rt, seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300], flush=True)
print("failures", bad)
