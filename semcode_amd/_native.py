"""ctypes binding of libsemcode_hip.so (include/semcode_hip.h).

This is the only place Python touches the C ABI.  There is no CPU fallback: if the shared
library is missing, or no MI355X is visible when a runtime is created, the error is raised
to the caller (the seams in embeddings/ and storage/ let it propagate as an ordinary
exception, which IndexerService / SemanticSearchPipeline already catch --
reference src/semcode/services/indexer.py:57-63, src/semcode/rag/pipeline.py:95-110).
"""
from __future__ import annotations

import ctypes as C
import sys
import threading
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libsemcode_hip.so"

METRICS = {"IP": 0, "L2": 1, "COSINE": 2}
KINDS = {"FLAT": 0, "IVF_FLAT": 1}


class ScError(RuntimeError):
    """A libsemcode_hip call returned a negative sc_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libsemcode_hip error {status}: {message}")
        self.status = status


class _Cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("stream", C.c_void_p), ("flags", C.c_int32)]


class EncoderCfg(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32),
                ("max_pos", C.c_int32), ("type_vocab", C.c_int32), ("ln_eps", C.c_float), ("normalize", C.c_int32),
                ("synth_seed", C.c_uint64), ("pos_type", C.c_int32), ("ffn_type", C.c_int32)]


_lib = None
_lib_lock = threading.Lock()

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes): exactly the entry points include/semcode_hip.h declares
SIGNATURES = {
    "sc_version": (C.c_char_p, []),
    "sc_last_error": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "sc_runtime_create": (C.c_int32, [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]),
    "sc_runtime_destroy": (C.c_int32, [C.c_void_p]),
    "sc_runtime_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "sc_runtime_synchronize": (C.c_int32, [C.c_void_p]),
    "sc_runtime_device_info": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "sc_runtime_set_profiling": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sc_runtime_profile_read": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "sc_runtime_profile_reset": (C.c_int32, [C.c_void_p]),
    "sc_synth_fill_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_int64]),
    "sc_encoder_blob_bytes": (C.c_int32, [C.POINTER(EncoderCfg), C.POINTER(C.c_int64)]),
    "sc_encoder_create": (C.c_int32, [C.c_void_p, C.POINTER(EncoderCfg), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sc_encoder_destroy": (C.c_int32, [C.c_void_p]),
    "sc_encoder_set_path": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sc_encoder_info": (C.c_int32, [C.c_void_p, C.POINTER(EncoderCfg)]),
    "sc_encoder_embed_ids": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sc_encoder_embed_ids_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sc_encoder_embed_ids_into": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sc_encoder_embed_ids_into_async": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_encoder_wait": (C.c_int32, [C.c_void_p]),
    "sc_tokenizer_create": (C.c_int32, [C.c_char_p, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]),
    "sc_tokenizer_destroy": (C.c_int32, [C.c_void_p]),
    "sc_tokenizer_info": (C.c_int32, [C.c_void_p] + [C.POINTER(C.c_int32)] * 5),
    "sc_tokenizer_encode": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "sc_diag_gemm_bf16": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sc_diag_gemm_bench": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "sc_diag_gemm_trace": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "sc_diag_set_option": (C.c_int32, [C.c_char_p, C.c_int32]),
    "sc_diag_gemm_i8": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sc_diag_encoder_read": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "sc_diag_attention": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sc_index_create": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "sc_index_destroy": (C.c_int32, [C.c_void_p]),
    "sc_index_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sc_index_reserve": (C.c_int32, [C.c_void_p, C.c_int64]),
    "sc_index_add": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sc_index_overwrite": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sc_index_put_rows": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sc_index_put_rows_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sc_index_get_rows": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "sc_index_fill_synthetic": (C.c_int32, [C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]),
    "sc_index_fill_synthetic_clustered": (C.c_int32, [C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_int32, C.c_float]),
    "sc_index_release_scratch": (C.c_int32, [C.c_void_p]),
    "sc_index_search": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_index_search_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_index_train": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64]),
    "sc_index_ivf_assignments": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "sc_index_set_ivf": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "sc_index_assign_lists": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "sc_index_ivf_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "sc_index_set_search_mode": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sc_index_last_search_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sc_index_set_coarse_stage": (C.c_int32, [C.c_void_p, C.c_int32]),
    "sc_index_last_coarse_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sc_index_last_collect_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sc_index_last_wide": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "sc_index_last_tail_rows": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    "sc_index_last_probe_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "sc_comm_unique_id": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "sc_comm_create": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sc_comm_destroy": (C.c_int32, [C.c_void_p]),
    "sc_comm_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sc_comm_rccl_version": (C.c_int32, [C.POINTER(C.c_int32)]),
    "sc_comm_allgather_topk": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_comm_broadcast": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32]),
    "sc_index_search_sharded": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_index_search_sharded_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sc_index_train_sharded": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "sc_comm_allreduce_max": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "sc_topk_merge_host": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python -m semcode_amd.csrc.build` "
                    "(needs hipcc); semcode_amd has no CPU fallback."
                )
            # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 under the system library's sonames, and the
            # copy that is loaded first serves the whole process.  With ours first, a later torch.cuda call in the same
            # process (torch.distributed in bench.py / storage.sharded, device tensors handed to the *_dev entry points)
            # fails with "No HIP GPUs are available"; so where torch is installed it is loaded first.  The library itself
            # never calls into torch.
            try:
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch is optional for the library
                pass
            handle = C.CDLL(str(LIB_PATH))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def _check(status: int) -> None:
    if status != 0:
        buf = C.create_string_buffer(512)
        lib().sc_last_error(buf, 512)
        raise ScError(status, buf.value.decode("utf-8", "replace"))


def _as_f32(a, shape_last: int | None = None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None and (arr.ndim != 2 or arr.shape[1] != shape_last):
        raise ValueError(f"expected a [n, {shape_last}] float array, got shape {arr.shape}")
    return arr


class Runtime:
    """One per process / GPU (sc_runtime)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        cfg = _Cfg(device=device, stream=C.c_void_p(stream) if stream else None, flags=0)
        _check(lib().sc_runtime_create(C.byref(cfg), C.byref(self._h)))
        self.device = device

    def close(self) -> None:
        if self._h:
            lib().sc_runtime_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover - best effort
        try:
            if not sys.is_finalizing():  # at interpreter shutdown handle order is arbitrary: leave it to process exit
                self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("runtime is closed")
        return self._h

    def set_stream(self, stream: int) -> None:
        _check(lib().sc_runtime_set_stream(self.handle, C.c_void_p(stream)))

    def synchronize(self) -> None:
        _check(lib().sc_runtime_synchronize(self.handle))

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int32(), C.c_int64()
        _check(lib().sc_runtime_device_info(self.handle, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}

    def set_profiling(self, enabled: "bool | int") -> None:
        """False / 0: off; True / 1: bracket every launch; n > 1: every n-th launch of the encoder kernel classes."""
        _check(lib().sc_runtime_set_profiling(self.handle, int(enabled)))

    def profile_read(self, which: int) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_int64()
        _check(lib().sc_runtime_profile_read(self.handle, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self) -> None:
        _check(lib().sc_runtime_profile_reset(self.handle))

    def synth_fill_dev(self, dev_ptr: int, rows: int, dim: int, ld: int, seed: int, first_row: int = 0) -> None:
        _check(lib().sc_synth_fill_dev(self.handle, C.c_void_p(dev_ptr), rows, dim, ld, seed, first_row))


_shared_runtimes: dict = {}
_shared_lock = threading.Lock()  # not _lib_lock: Runtime() takes that one inside lib()


def shared_runtime(device: int = 0) -> Runtime:
    """The process-wide runtime of a device.  The embedding client and the vector store use it by default, so that the
    encoder's output buffer and the index live on one device and one stream and a batch can go from one to the other
    without touching host memory (Encoder.embed_ids_into).  Never closed explicitly: it lives as long as the process."""
    with _shared_lock:
        rt = _shared_runtimes.get(int(device))
        if rt is None or not rt._h:
            rt = Runtime(device=int(device))
            _shared_runtimes[int(device)] = rt
        return rt


class Index:
    """HBM-resident vector index (sc_index): FLAT or IVF_FLAT, metric IP / L2 / COSINE."""

    def __init__(self, rt: Runtime, dim: int, metric: str = "IP", kind: str = "FLAT", nlist: int = 128, row_base: int = 0):
        if metric not in METRICS:
            raise ValueError(f"unknown metric {metric!r}")
        if kind not in KINDS:
            raise ValueError(f"unknown index kind {kind!r}")
        self.rt = rt
        self.dim = int(dim)
        self.metric = metric
        self.kind = kind
        self.row_base = int(row_base)
        self._h = C.c_void_p()
        _check(lib().sc_index_create(rt.handle, self.dim, METRICS[metric], KINDS[kind], int(nlist), self.row_base, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().sc_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("index is closed")
        return self._h

    def info(self) -> dict:
        rows, dim, ld = C.c_int64(), C.c_int32(), C.c_int32()
        _check(lib().sc_index_info(self.handle, C.byref(rows), C.byref(dim), C.byref(ld)))
        return {"rows": rows.value, "dim": dim.value, "ld": ld.value}

    def __len__(self) -> int:
        return self.info()["rows"]

    def reserve(self, rows: int) -> None:
        _check(lib().sc_index_reserve(self.handle, int(rows)))

    def add(self, vecs) -> None:
        v = _as_f32(vecs, self.dim)
        _check(lib().sc_index_add(self.handle, v.ctypes.data_as(C.c_void_p), v.shape[0]))

    def overwrite(self, vecs, rows) -> None:
        v = _as_f32(vecs, self.dim)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        if r.shape != (v.shape[0],):
            raise ValueError("rows must have one entry per vector")
        _check(lib().sc_index_overwrite(self.handle, v.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), v.shape[0]))

    def put_rows(self, vecs, rows) -> None:
        """rows[i] <- vecs[i]; a row number is an existing row (replace) or the next free one (append)."""
        v = _as_f32(vecs, self.dim)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        if r.shape != (v.shape[0],):
            raise ValueError("rows must have one entry per vector")
        _check(lib().sc_index_put_rows(self.handle, v.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), v.shape[0]))

    def put_rows_dev(self, vecs_ptr: int, rows) -> None:
        """Same with a DEVICE pointer to tight [n, dim] f32 vectors (e.g. a torch tensor's data_ptr())."""
        r = np.ascontiguousarray(rows, dtype=np.int64)
        _check(lib().sc_index_put_rows_dev(self.handle, C.c_void_p(int(vecs_ptr)), r.ctypes.data_as(C.c_void_p), r.shape[0]))

    def get_rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), dtype=np.float32)
        _check(lib().sc_index_get_rows(self.handle, int(first), int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def fill_synthetic(self, n: int, seed: int, first_row: int = 0) -> None:
        _check(lib().sc_index_fill_synthetic(self.handle, int(n), int(seed), int(first_row)))

    def fill_synthetic_clustered(self, n: int, seed: int, nclusters: int, spread: float, first_row: int = 0) -> None:
        _check(lib().sc_index_fill_synthetic_clustered(self.handle, int(n), int(seed), int(first_row), int(nclusters), float(spread)))

    def release_scratch(self) -> None:
        _check(lib().sc_index_release_scratch(self.handle))

    def search(self, queries, k: int = 10, nprobe: int = 16) -> tuple[np.ndarray, np.ndarray]:
        """queries [Q, dim] -> (dist [Q, k] f32, rows [Q, k] i64), best first."""
        q = _as_f32(queries, self.dim)
        Q = q.shape[0]
        dist = np.empty((Q, k), dtype=np.float32)
        rows = np.empty((Q, k), dtype=np.int64)
        _check(lib().sc_index_search(self.handle, q.ctypes.data_as(C.c_void_p), Q, int(k), int(nprobe),
                                     dist.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p)))
        return dist, rows

    def train(self, niter: int = 10, seed: int = 0) -> None:
        """IVF_FLAT: k-means + list build (sc_index_train)."""
        _check(lib().sc_index_train(self.handle, int(niter), int(seed)))

    def ivf_info(self) -> dict:
        n = C.c_int32()
        _check(lib().sc_index_ivf_info(self.handle, C.byref(n), None, None))
        if n.value == 0:
            return {"nlist": 0}
        cent = np.empty((n.value, self.dim), dtype=np.float32)
        sizes = np.empty((n.value,), dtype=np.int64)
        _check(lib().sc_index_ivf_info(self.handle, C.byref(n), cent.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p)))
        return {"nlist": n.value, "centroids": cent, "list_sizes": sizes}

    def ivf_assignments(self) -> np.ndarray:
        """List of every row, insertion order (int32 [rows]); the index must be trained."""
        out = np.empty(len(self), dtype=np.int32)
        _check(lib().sc_index_ivf_assignments(self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_ivf(self, centroids, assign) -> None:
        """Install a saved IVF structure (centroids [nlist, dim], list of every row) without re-running k-means."""
        c = _as_f32(centroids, self.dim)
        a = np.ascontiguousarray(assign, dtype=np.int32)
        if a.shape != (len(self),):
            raise ValueError("assign must have one entry per stored row")
        _check(lib().sc_index_set_ivf(self.handle, c.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), c.shape[0]))

    def assign_lists(self, centroids) -> None:
        """Build the IVF lists for given centroids [nlist, dim] (no k-means): what every rank of a sharded collection does
        with the centroids one rank trained."""
        c = _as_f32(centroids, self.dim)
        _check(lib().sc_index_assign_lists(self.handle, c.ctypes.data_as(C.c_void_p), c.shape[0]))

    def set_search_mode(self, mode: str) -> None:
        """'auto' | 'exact' | 'batched' | 'ivf' (per-query probing) | 'ivf_listmajor' (exact f32 list-major probing) | 'ivf_coarse' (list-major probing
        behind the int8 coarse stage; L2 only) -- see sc_index_set_search_mode."""
        _check(lib().sc_index_set_search_mode(self.handle, {"auto": 0, "exact": 1, "batched": 2, "ivf": 3, "ivf_listmajor": 4, "ivf_coarse": 5}[mode]))

    def set_coarse_stage(self, bits: int) -> None:
        """First coarse stage of the batched path: 0 auto (int8, then bf16), 8 int8 only, 16 bf16 only (sc_index_set_coarse_stage)."""
        _check(lib().sc_index_set_coarse_stage(self.handle, int(bits)))

    def last_search_stats(self) -> dict:
        """path; `uncertified` = queries that ended in the exact scan; for the batched path also the stage it started on
        (`coarse_bits` 8 / 16), how many queries the int8 stage handed to the bf16 stage (`handed_to_bf16`), and how many
        queries went through a collect pass after a failed certificate / were answered by it (`collect_tried`, `collect_resolved`)."""
        path, unc, bits, handed = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().sc_index_last_search_stats(self.handle, C.byref(path), C.byref(unc)))
        _check(lib().sc_index_last_coarse_stats(self.handle, C.byref(bits), C.byref(handed)))
        out = {"path": {0: "none", 1: "exact", 2: "batched", 3: "ivf", 4: "ivf_listmajor", 5: "ivf_coarse"}[path.value], "uncertified": unc.value}
        if path.value in (3, 4, 5):
            tail = C.c_int64()
            _check(lib().sc_index_last_tail_rows(self.handle, C.byref(tail)))
            out["tail_rows"] = int(tail.value)
        if path.value == 2:
            tried, resolved = C.c_int32(), C.c_int32()
            _check(lib().sc_index_last_collect_stats(self.handle, C.byref(tried), C.byref(resolved)))
            wide = C.c_int32()
            _check(lib().sc_index_last_wide(self.handle, C.byref(wide)))
            out.update(coarse_bits=bits.value, handed_to_bf16=handed.value, collect_tried=tried.value, collect_resolved=resolved.value, wide=bool(wide.value))
        return out

    def search_sharded(self, comm: "Comm", queries, k: int = 10, nprobe: int = 16) -> tuple[np.ndarray, np.ndarray]:
        """Row-sharded search (every rank calls it with the same queries): -> the merged global (dist, rows) [Q, k]."""
        q = _as_f32(queries, self.dim)
        Q = q.shape[0]
        dist = np.empty((Q, k), dtype=np.float32)
        rows = np.empty((Q, k), dtype=np.int64)
        _check(lib().sc_index_search_sharded(self.handle, comm.handle, q.ctypes.data_as(C.c_void_p), Q, int(k), int(nprobe),
                                             dist.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p)))
        return dist, rows

    def search_sharded_dev(self, comm: "Comm", q_ptr: int, Q: int, k: int, all_dist_ptr: int, all_rows_ptr: int, nprobe: int = 16) -> None:
        """Device-pointer variant: every shard's [Q, k] into all_dist / all_rows [world, Q, k] (sc_index_search_sharded_dev)."""
        _check(lib().sc_index_search_sharded_dev(self.handle, comm.handle, C.c_void_p(q_ptr), int(Q), int(k), int(nprobe),
                                                 C.c_void_p(all_dist_ptr), C.c_void_p(all_rows_ptr)))

    def train_sharded(self, comm: "Comm", niter: int = 10, root: int = 0) -> None:
        _check(lib().sc_index_train_sharded(self.handle, comm.handle, int(niter), int(root)))

    def last_probe_stats(self) -> dict:
        """After a list-major IVF probe: rows of the distinct probed lists, rows streamed, work items (sc_index_last_probe_stats)."""
        u, st, g = C.c_int64(), C.c_int64(), C.c_int32()
        _check(lib().sc_index_last_probe_stats(self.handle, C.byref(u), C.byref(st), C.byref(g)))
        return {"unique_rows": u.value, "streamed_rows": st.value, "groups": g.value}

    def search_dev(self, q_ptr: int, Q: int, k: int, dist_ptr: int, rows_ptr: int, nprobe: int = 16) -> None:
        """Device-pointer variant (enqueued on the runtime's stream; see sc_index_search_dev for where it synchronises)."""
        _check(lib().sc_index_search_dev(self.handle, C.c_void_p(q_ptr), int(Q), int(k), int(nprobe), C.c_void_p(dist_ptr), C.c_void_p(rows_ptr)))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """The RCCL rendezvous id (rank 0 creates it and hands the 128 bytes to the other ranks out of band)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().sc_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class Comm:
    """RCCL communicator of the path's two collectives (sc_comm): the all-gather of per-shard top-k and the centroid broadcast.
    Device pointers in, device pointers out, everything on the runtime's stream."""

    def __init__(self, rt: Runtime, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        self.rt, self.rank, self.world = rt, int(rank), int(world)
        self._h = C.c_void_p()
        _check(lib().sc_comm_create(rt.handle, self.rank, self.world, unique_id, COMM_ID_BYTES, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().sc_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("communicator is closed")
        return self._h

    def info(self) -> dict:
        """What the communicator itself reports (sc_comm_info, sc_comm_rccl_version): rank, world, RCCL version."""
        r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().sc_comm_info(self.handle, C.byref(r), C.byref(w)))
        _check(lib().sc_comm_rccl_version(C.byref(v)))
        return {"rank": r.value, "world": w.value, "rccl_version": v.value}

    def allgather_topk(self, dist_ptr: int, rows_ptr: int, Q: int, k: int, all_dist_ptr: int, all_rows_ptr: int) -> None:
        _check(lib().sc_comm_allgather_topk(self.handle, C.c_void_p(dist_ptr), C.c_void_p(rows_ptr), int(Q), int(k),
                                            C.c_void_p(all_dist_ptr), C.c_void_p(all_rows_ptr)))

    def broadcast(self, buf_ptr: int, nbytes: int, root: int = 0) -> None:
        _check(lib().sc_comm_broadcast(self.handle, C.c_void_p(buf_ptr), int(nbytes), int(root)))

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        _check(lib().sc_comm_allreduce_max(self.handle, C.byref(v)))
        return v.value


BERT_BASE = dict(vocab=30522, hidden=768, layers=12, heads=12, ffn=3072, max_pos=512, type_vocab=2, ln_eps=1e-12)
SEQ_BUCKETS = (32, 64, 128, 256, 512, 1024, 2048)  # > 512: position-free (ALiBi) encoders, or a position table that long


class Encoder:
    """Transformer-encoder forward on the device (sc_encoder): token ids in, pooled f32 vectors out."""

    def __init__(self, rt: Runtime, cfg: dict | None = None, weights: np.ndarray | None = None, normalize: bool = False,
                 synth_seed: int = 0):
        c = dict(BERT_BASE)
        c.update(cfg or {})
        self.cfg = EncoderCfg(vocab=c["vocab"], hidden=c["hidden"], layers=c["layers"], heads=c["heads"], ffn=c["ffn"],
                              max_pos=c["max_pos"], type_vocab=c["type_vocab"], ln_eps=c["ln_eps"], normalize=1 if normalize else 0,
                              synth_seed=synth_seed, pos_type=1 if c.get("alibi") else 0, ffn_type=1 if c.get("geglu") else 0)
        self.rt = rt
        self.hidden = c["hidden"]
        self.max_pos = c["max_pos"]
        self._h = C.c_void_p()
        need = C.c_int64()
        _check(lib().sc_encoder_blob_bytes(C.byref(self.cfg), C.byref(need)))
        self.blob_bytes = need.value
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
            _check(lib().sc_encoder_create(rt.handle, C.byref(self.cfg), w.ctypes.data_as(C.c_void_p), w.nbytes, C.byref(self._h)))
        else:
            _check(lib().sc_encoder_create(rt.handle, C.byref(self.cfg), None, 0, C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().sc_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("encoder is closed")
        return self._h

    def set_path(self, path: str) -> None:
        """'auto' | 'batch' (LayerNorm-folded 256-tile pipeline for every size) | 'small' (split-K + LayerNorm kernels)."""
        _check(lib().sc_encoder_set_path(self.handle, {"auto": 0, "batch": 1, "small": 2}[path]))

    def embed_ids(self, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
        """ids [B, S] int32 with S in SEQ_BUCKETS, lens [B] -> [B, hidden] f32."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if ids.ndim != 2 or lens.shape != (ids.shape[0],):
            raise ValueError("ids must be [B, S] and lens [B]")
        B, S = ids.shape
        out = np.empty((B, self.hidden), dtype=np.float32)
        _check(lib().sc_encoder_embed_ids(self.handle, ids.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, S,
                                          out.ctypes.data_as(C.c_void_p)))
        return out

    def embed_ids_into(self, ids: np.ndarray, lens: np.ndarray, index: "Index", rows: np.ndarray, want_host: bool = False,
                       wait: bool = True) -> "np.ndarray | None":
        """Embed one batch and store the vectors in `index` rows `rows` (existing rows are replaced, the next free rows
        appended) without a trip through host memory; returns the vectors only if want_host.  wait=False: return as soon as
        the batch is enqueued (at most two in flight; call wait() before relying on completion)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if ids.ndim != 2 or lens.shape != (ids.shape[0],) or rows.shape != (ids.shape[0],):
            raise ValueError("ids must be [B, S], lens [B] and rows [B]")
        B, S = ids.shape
        if not wait and not want_host:
            _check(lib().sc_encoder_embed_ids_into_async(self.handle, ids.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, S,
                                                         index.handle, rows.ctypes.data_as(C.c_void_p)))
            return None
        out = np.empty((B, self.hidden), dtype=np.float32) if want_host else None
        _check(lib().sc_encoder_embed_ids_into(self.handle, ids.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, S, index.handle,
                                               rows.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p) if want_host else None))
        return out

    def wait(self) -> None:
        """Block until every batch enqueued with embed_ids_into(..., wait=False) has finished."""
        _check(lib().sc_encoder_wait(self.handle))

    def embed_ids_dev(self, ids_ptr: int, lens_ptr: int, B: int, S: int, out_ptr: int) -> None:
        _check(lib().sc_encoder_embed_ids_dev(self.handle, C.c_void_p(ids_ptr), C.c_void_p(lens_ptr), int(B), int(S), C.c_void_p(out_ptr)))


class NativeTokenizer:
    """C++ WordPiece (ASCII fast path + BERT's Unicode normalisation, multi-threaded).  encode_batch also returns which texts it
    left alone (malformed UTF-8 only)."""

    def __init__(self, vocab_path, lowercase: bool = True):
        data = Path(vocab_path).read_bytes()
        self._h = C.c_void_p()
        _check(lib().sc_tokenizer_create(data, len(data), 1 if lowercase else 0, C.byref(self._h)))
        v = [C.c_int32() for _ in range(5)]
        _check(lib().sc_tokenizer_info(self._h, *[C.byref(x) for x in v]))
        self.vocab_size, self.pad_id, self.unk_id, self.cls_id, self.sep_id = (x.value for x in v)

    def close(self) -> None:
        if self._h:
            lib().sc_tokenizer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def encode_batch(self, texts, max_tokens: int, S: int, threads: int = 0):
        """-> (ids [n,S] int32, lens [n] int32, needs_fallback [n] bool)"""
        raw = [t.encode("utf-8") for t in texts]
        offsets = np.zeros(len(raw) + 1, dtype=np.int64)
        np.cumsum([len(r) for r in raw], out=offsets[1:])
        blob = b"".join(raw)
        n = len(raw)
        ids = np.empty((n, S), dtype=np.int32)
        lens = np.empty((n,), dtype=np.int32)
        fb = np.empty((n,), dtype=np.uint8)
        _check(lib().sc_tokenizer_encode(self._h, blob, offsets.ctypes.data_as(C.c_void_p), n, int(max_tokens), int(S),
                                         ids.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p), int(threads)))
        return ids, lens, fb.astype(bool)


def diag_gemm_bf16(rt: Runtime, A, W, bias, R=None, epi: int = 0) -> np.ndarray:
    """One GEMM kernel launch on host data (parity tests): out [M,N] = A [M,K] @ W [N,K].T (+ epilogue)."""
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32)
    Rp = None
    if R is not None:
        R = np.ascontiguousarray(R, np.float32)
        Rp = R.ctypes.data_as(C.c_void_p)
    _check(lib().sc_diag_gemm_bf16(rt.handle, epi, A.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
                                   bias.ctypes.data_as(C.c_void_p), Rp, M, N, K, out.ctypes.data_as(C.c_void_p)))
    return out


def diag_set_option(name: str, value: int) -> None:
    """Process-wide test / tuning switch of the library (sc_diag_set_option)."""
    _check(lib().sc_diag_set_option(name.encode(), int(value)))


def diag_gemm_bench(rt: Runtime, M: int, N: int, K: int, epi: int = 0, iters: int = 20, variant: int = 0) -> float:
    """ms per launch of one GEMM shape on device-resident synthetic data (tuning aid)."""
    ms = C.c_double()
    _check(lib().sc_diag_gemm_bench(rt.handle, epi, M, N, K, iters, variant, C.byref(ms)))
    return ms.value


def diag_gemm_trace(rt: Runtime, M: int, N: int, K: int, epi: int = 0, launches: int = 1) -> np.ndarray:
    """Per-workgroup time stamps of back-to-back 256-tile GEMM launches: [launches, ntiles, 8] uint64 (see include/semcode_hip.h)."""
    nt = (M // 256) * (N // 256)
    out = np.zeros((launches, nt, 8), np.uint64)
    _check(lib().sc_diag_gemm_trace(rt.handle, epi, M, N, K, out.ctypes.data_as(C.c_void_p), out.size))
    return out


def diag_gemm_i8(rt: Runtime, A, W) -> np.ndarray:
    """out [M,N] int32 = A [M,K] int8 @ W [N,K].T on the int8 form of the 256 tile (exact)."""
    A = np.ascontiguousarray(A, np.int8)
    W = np.ascontiguousarray(W, np.int8)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.int32)
    _check(lib().sc_diag_gemm_i8(rt.handle, A.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), M, N, K, out.ctypes.data_as(C.c_void_p)))
    return out


def diag_attention(rt: Runtime, qkv, lens, B: int, S: int, heads: int) -> np.ndarray:
    qkv = np.ascontiguousarray(qkv, np.float32)
    lens = np.ascontiguousarray(lens, np.int32)
    out = np.empty((B * S, heads * 64), np.float32)
    _check(lib().sc_diag_attention(rt.handle, qkv.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, S, heads,
                                   out.ctypes.data_as(C.c_void_p)))
    return out


def topk_merge_host(metric: str, dist: np.ndarray, rows: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Merge per-shard results dist/rows [lists, Q, k] -> global best-first [Q, k]."""
    d = np.ascontiguousarray(dist, dtype=np.float32)
    r = np.ascontiguousarray(rows, dtype=np.int64)
    if d.ndim != 3 or d.shape != r.shape:
        raise ValueError("dist and rows must both be [lists, Q, k]")
    lists, Q, k = d.shape
    od = np.empty((Q, k), dtype=np.float32)
    orow = np.empty((Q, k), dtype=np.int64)
    _check(lib().sc_topk_merge_host(METRICS[metric], lists, Q, k, d.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                    od.ctypes.data_as(C.c_void_p), orow.ctypes.data_as(C.c_void_p)))
    return od, orow
