"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under semcode_amd/ does.  It wraps oracle/sc_oracle.c (built into oracle/_build/libsc_oracle.so by
`build()`), and adds an independent numpy float64 brute force used to pin the C restatement.

Parity status: "parity unpinned" at the third-party boundary (see the header of sc_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "sc_oracle.c"
LIB = HERE / "_build" / "libsc_oracle.so"

METRICS = {"IP": 0, "L2": 1, "COSINE": 2}
_lib = None


def build(force: bool = False) -> Path:
    """gcc -O2 -mfma -fopenmp; x86-64-v3 only (no -march=native: the .so travels to the GPU box)."""
    if not force and LIB.exists() and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return LIB
    gcc = shutil.which("gcc")
    if not gcc:
        raise RuntimeError("gcc not found: cannot build the oracle")
    LIB.parent.mkdir(exist_ok=True)
    cmd = [gcc, "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
           str(SRC), "-o", str(LIB), "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        h = C.CDLL(str(LIB))
        h.sc_oracle_synth.restype = C.c_float
        h.sc_oracle_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        h.sc_oracle_synth_fill.restype = None
        h.sc_oracle_synth_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_int64]
        h.sc_oracle_sqnorm.restype = C.c_float
        h.sc_oracle_sqnorm.argtypes = [C.c_void_p, C.c_int32]
        h.sc_oracle_dot.restype = C.c_float
        h.sc_oracle_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        h.sc_oracle_score.restype = C.c_float
        h.sc_oracle_score.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float]
        h.sc_oracle_search.restype = None
        h.sc_oracle_search.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                       C.c_void_p, C.c_void_p]
        h.sc_oracle_search_rows.restype = None
        h.sc_oracle_search_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                            C.c_void_p, C.c_void_p]
        h.sc_oracle_threads.restype = C.c_int32
        h.sc_oracle_synth_rows.restype = None
        h.sc_oracle_synth_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_uint64]
        h.sc_oracle_centroid_mean.restype = None
        h.sc_oracle_centroid_mean.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        h.sc_oracle_set_threads.restype = None
        h.sc_oracle_set_threads.argtypes = [C.c_int32]
        # never more OpenMP threads than CPUs this process may run on (cgroup-limited GPU boxes)
        try:
            h.sc_oracle_set_threads(host_cores())
        except Exception:  # pragma: no cover
            pass
        _lib = h
    return _lib


def host_cores() -> int:
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box gives a job a share
    of its cores; starting one thread per visible core on a 16-core quota runs 10x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p_ + 0.5)))
        except Exception:
            pass
    return max(1, n)


def pad_ld(dim: int, align: int = 64) -> int:
    return (dim + align - 1) // align * align


def padded(a: np.ndarray, ld: int | None = None) -> np.ndarray:
    """[n, dim] -> zero-padded contiguous [n, ld] float32 (the HBM row layout)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    ld = ld or pad_ld(a.shape[1])
    if a.shape[1] == ld:
        return a
    out = np.zeros((a.shape[0], ld), dtype=np.float32)
    out[:, : a.shape[1]] = a
    return out


def synth(rows: int, dim: int, seed: int, first_row: int = 0, ld: int | None = None) -> np.ndarray:
    """The deterministic synthetic generator (bit-identical to the device kernel)."""
    ld = ld or dim
    out = np.empty((rows, ld), dtype=np.float32)
    lib().sc_oracle_synth_fill(out.ctypes.data_as(C.c_void_p), rows, dim, ld, seed, first_row)
    return out


def synth_rows(rows_idx, dim: int, seed: int) -> np.ndarray:
    """Regenerate arbitrary (global) rows of the synthetic corpus: random access, no storage."""
    rows_idx = np.ascontiguousarray(rows_idx, dtype=np.int64)
    out = np.empty((len(rows_idx), dim), dtype=np.float32)
    lib().sc_oracle_synth_rows(out.ctypes.data_as(C.c_void_p), rows_idx.ctypes.data_as(C.c_void_p), len(rows_idx), dim, seed)
    return out


M64 = (1 << 64) - 1


def _mix64(z: int) -> int:
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_clustered(rows: int, dim: int, seed: int, nclusters: int, spread: float, first_row: int = 0) -> np.ndarray:
    """Clustered corpus of sc_index_fill_synthetic_clustered: centre[hash(row) % nclusters] + spread * noise (one fmaf)."""
    cseed = seed ^ 0xC1057E25
    ckey = _mix64((cseed + 0x9E3779B97F4A7C15) & M64)
    cl = np.array([_mix64(ckey ^ (((first_row + r) * 0x9E3779B97F4A7C15) & M64)) % nclusters for r in range(rows)], dtype=np.int64)
    noise = synth(rows, dim, seed, first_row=first_row)
    centres = synth_rows(cl, dim, cseed)
    # fmaf(spread, noise, centre): one rounding -> do it in float64 (exact product of two f32 fits) and round once
    return (np.float64(np.float32(spread)) * noise.astype(np.float64) + centres.astype(np.float64)).astype(np.float32)


def synth_clustered_rows(rows_idx, dim: int, seed: int, nclusters: int, spread: float) -> np.ndarray:
    """Arbitrary (global) rows of the clustered corpus: the same rule as synth_clustered, random access."""
    rows_idx = np.ascontiguousarray(rows_idx, dtype=np.int64)
    cseed = seed ^ 0xC1057E25
    ckey = _mix64((cseed + 0x9E3779B97F4A7C15) & M64)
    cl = np.array([_mix64(ckey ^ ((int(r) * 0x9E3779B97F4A7C15) & M64)) % nclusters for r in rows_idx], dtype=np.int64)
    noise = synth_rows(rows_idx, dim, seed)
    centres = synth_rows(cl, dim, cseed)
    return (np.float64(np.float32(spread)) * noise.astype(np.float64) + centres.astype(np.float64)).astype(np.float32)


def sqnorm(x: np.ndarray) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().sc_oracle_sqnorm(x.ctypes.data_as(C.c_void_p), x.shape[0]))


def dot(x: np.ndarray, q: np.ndarray) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    assert x.shape == q.shape and x.shape[0] % 16 == 0
    return float(lib().sc_oracle_dot(x.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), x.shape[0]))


def search(X: np.ndarray, Q: np.ndarray, k: int, metric: str = "L2", row_base: int = 0):
    """Canonical-order exhaustive search.  X [n, dim], Q [nq, dim] -> (dist [nq,k] f32, rows [nq,k] i64)."""
    Xp, Qp = padded(X), padded(Q)
    assert Xp.shape[1] == Qp.shape[1]
    nq = Qp.shape[0]
    dist = np.empty((nq, k), dtype=np.float32)
    rows = np.empty((nq, k), dtype=np.int64)
    lib().sc_oracle_search(Xp.ctypes.data_as(C.c_void_p), Xp.shape[0], Xp.shape[1], METRICS[metric],
                           Qp.ctypes.data_as(C.c_void_p), nq, k, row_base, dist.ctypes.data_as(C.c_void_p),
                           rows.ctypes.data_as(C.c_void_p))
    return dist, rows


def search_rows(X: np.ndarray, q: np.ndarray, rows, k: int, metric: str = "L2"):
    """Canonical-order search restricted to `rows` of X for one query."""
    Xp, qp = padded(X), padded(q.reshape(1, -1))
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    dist = np.empty((k,), dtype=np.float32)
    out = np.empty((k,), dtype=np.int64)
    lib().sc_oracle_search_rows(Xp.ctypes.data_as(C.c_void_p), Xp.shape[1], METRICS[metric], qp.ctypes.data_as(C.c_void_p),
                                rows.ctypes.data_as(C.c_void_p), len(rows), k, dist.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p))
    return dist, out


def threads() -> int:
    return int(lib().sc_oracle_threads())


def search_sgemm(X: np.ndarray, Q: np.ndarray, k: int, metric: str = "L2", threads: int | None = None, block: int = 131072):
    """The formulation a CPU implementation would use (bench.py's cpu_baseline): scores by BLAS sgemm over blocks of rows
    (L2 = |x|^2 + |q|^2 - 2 X Q^T), a partial sort per block, a final (score, row) sort of the block winners.  f32 arithmetic
    in BLAS order, so distances agree with search() to rounding only -- a timing twin, not the parity oracle."""
    import torch

    if threads:
        torch.set_num_threads(int(threads))
    Xt, Qt = torch.from_numpy(np.ascontiguousarray(X, np.float32)), torch.from_numpy(np.ascontiguousarray(Q, np.float32))
    qn = (Qt * Qt).sum(1)
    best_s, best_r = [], []
    with torch.no_grad():
        for r0 in range(0, Xt.shape[0], block):
            xb = Xt[r0:r0 + block]
            s = Qt @ xb.T
            if metric == "L2":
                s = (xb * xb).sum(1)[None, :] + qn[:, None] - 2.0 * s
            elif metric == "COSINE":
                s = s / (torch.sqrt((xb * xb).sum(1))[None, :] * torch.sqrt(qn)[:, None])
            v, i = torch.topk(s, min(k, xb.shape[0]), dim=1, largest=(metric != "L2"))
            best_s.append(v)
            best_r.append(i + r0)
        v, r = torch.cat(best_s, 1).numpy(), torch.cat(best_r, 1).numpy()
    key = v if metric == "L2" else -v
    order = np.lexsort((r, key), axis=1)[:, :k]
    return np.take_along_axis(v, order, 1).astype(np.float32), np.take_along_axis(r, order, 1).astype(np.int64)


# ---------------------------------------------------------------- independent float64 reference

def search_f64(X: np.ndarray, Q: np.ndarray, k: int, metric: str = "L2"):
    """numpy float64 brute force, tie rule (score, lower row).  Returns (score f64 [nq,k], rows i64 [nq,k])."""
    X64 = np.asarray(X, dtype=np.float64)
    Q64 = np.asarray(Q, dtype=np.float64)
    dots = Q64 @ X64.T
    if metric == "L2":
        s = (X64 * X64).sum(1)[None, :] + (Q64 * Q64).sum(1)[:, None] - 2.0 * dots
        order_key = s
    elif metric == "COSINE":
        s = dots / (np.sqrt((X64 * X64).sum(1))[None, :] * np.sqrt((Q64 * Q64).sum(1))[:, None])
        order_key = -s
    else:
        s = dots
        order_key = -s
    n = X64.shape[0]
    kk = min(k, n)
    rows = np.full((Q64.shape[0], k), -1, dtype=np.int64)
    score = np.full((Q64.shape[0], k), np.inf if metric == "L2" else -np.inf)
    for i in range(Q64.shape[0]):
        idx = np.lexsort((np.arange(n), order_key[i]))[:kk]
        rows[i, :kk] = idx
        score[i, :kk] = s[i, idx]
    return score, rows
