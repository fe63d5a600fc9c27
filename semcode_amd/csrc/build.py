"""Build libsemcode_hip.so (gfx950) in-tree with hipcc.

    python -m semcode_amd.csrc.build [--force] [--save-temps]

Objects are rebuilt only when a source or header is newer.  The .so lands in
semcode_amd/_lib/libsemcode_hip.so (git-ignored, travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
ROOT = CSRC.parent.parent
OUT_DIR = CSRC.parent / "_lib"
OBJ_DIR = CSRC / "_obj"
LIB = OUT_DIR / "libsemcode_hip.so"

HIP_SOURCES = sorted(p.name for p in CSRC.glob("*.hip"))
CPP_SOURCES = sorted(p.name for p in CSRC.glob("*.cpp"))

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", f"-I{ROOT / 'include'}", "-I/opt/rocm/include"]
HIP_FLAGS = ["--offload-arch=gfx950", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; libsemcode_hip.so cannot be built")
    return exe


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OUT_DIR.mkdir(exist_ok=True)
    OBJ_DIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h"))
    jobs = []
    objs = []
    for name in HIP_SOURCES + CPP_SOURCES:
        src = CSRC / name
        obj = OBJ_DIR / (name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc, *COMMON]
            if name.endswith(".hip"):
                cmd += HIP_FLAGS
                if save_temps:
                    cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            else:
                cmd += ["-x", "hip", "--offload-arch=gfx950"]
            cmd += ["-c", str(src), "-o", str(obj)]
            jobs.append((name, cmd))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(OBJ_DIR))
        return name, r

    failed = False
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for name, r in ex.map(run, jobs):
            if verbose and (r.stderr.strip() or r.returncode):
                sys.stderr.write(f"--- {name}\n{r.stderr}\n")
            if r.returncode:
                failed = True
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    if jobs or force or not LIB.exists():
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *map(str, objs), "-ldl", "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link of libsemcode_hip.so failed")
    if verbose:
        print(f"built {LIB} ({len(jobs)} object(s) recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
