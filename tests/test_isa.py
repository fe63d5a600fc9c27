"""Static check of the built gfx950 code objects (CPU only: disassembly of libsemcode_hip.so).

On MI355X, `v_pk_mul_f32 D, S0, S1 op_sel:[0,1]` (a packed-f32 op whose LOW result lane takes the HIGH register of an operand)
intermittently produced 0 in lanes 48..63 while other waves of the CU were inside their MFMA loop -- the cause of the one
non-reproducible encoder output this code base ever had (gemm_bf16.hip SC_OPAQUE_PAIR).  hipcc picks that
encoding by itself whenever a broadcast scalar sits in the odd register of a pair, so the kernels that run MFMAs are checked here
for it after every build.  Round 3: the failure was observed while OTHER waves of the CU were inside their MFMA loop, so a kernel
without MFMAs of its own is exposed too once two streams share the device (two runtimes in one process): every kernel of the
library is checked, not only the ones that contain MFMAs."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

from semcode_amd import _native

OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")


@pytest.mark.skipif(not OBJDUMP.exists(), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_low_from_high_operand_select_on_packed_f32_in_mfma_kernels(tmp_path):
    lib = tmp_path / "lib.so"
    shutil.copy(_native.LIB_PATH, lib)
    subprocess.run([str(OBJDUMP), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)  # writes lib.so.<n>.<target>
    objs = sorted(tmp_path.glob("lib.so.*gfx950"))
    assert objs, "no gfx950 code object in libsemcode_hip.so"
    bad, mfma_kernels = [], 0
    for co in objs:
        dis = subprocess.run([str(OBJDUMP), "-d", str(co)], check=True, capture_output=True, text=True).stdout
        kernel, lines, has_mfma = None, [], False

        def close():
            nonlocal mfma_kernels
            if kernel and has_mfma:
                mfma_kernels += 1
            if kernel:  # EVERY kernel: a LayerNorm / pooling / shadow kernel can be co-resident with another stream's MFMA kernel
                bad.extend((kernel, l.strip()) for l in lines)

        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                close()
                kernel, lines, has_mfma = m.group(1), [], False
            elif "v_mfma" in line:
                has_mfma = True
            elif re.search(r"v_pk_\w+_f32\b.*\bop_sel:\[", line):
                lines.append(line)
        close()
    assert mfma_kernels >= 8  # the GEMM, attention and coarse-scan kernels were seen
    assert not bad, f"{len(bad)} packed-f32 ops with op_sel (low lane from a high register), e.g. {bad[:3]}"
