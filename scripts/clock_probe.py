"""Does the GEMM run power-limited?  Samples rocm-smi (sclk, power) while a GEMM shape loops (tuning aid).
    python3 scripts/clock_probe.py [variant]   (variant 0 = full kernel, 21 = MFMAs only, 4 = no epilogue)"""
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
from semcode_amd import _native

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rt = _native.Runtime(0)
stop = False
res = []


def work():
    while not stop:
        res.append(_native.diag_gemm_bench(rt, 65536, 2304, 768, epi=0, iters=200, variant=variant))


t = threading.Thread(target=work)
t.start()
time.sleep(1.0)
for _ in range(4):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "fclk" in l or "mclk" in l) and "GPU[0]" in l]
    print(" | ".join(keep), flush=True)
    time.sleep(0.7)
stop = True
t.join()
ms = sum(res) / len(res)
print(f"variant {variant}: {ms*1e3:.1f} us per launch = {2.0*65536*2304*768/ms/1e9:.0f} TF")
