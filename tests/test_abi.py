"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def built():
    from semcode_amd.csrc import build

    return build.build(verbose=False)


def test_header_declares_something():
    syms = declared_symbols()
    assert "sc_index_search" in syms and "sc_runtime_create" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(str(built))
    missing = [s for s in declared_symbols() if not hasattr(handle, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_binding_covers_every_declared_symbol(built):
    from semcode_amd import _native

    assert sorted(_native.SIGNATURES) == declared_symbols()
    _native.lib()  # resolves all of them with argtypes
    assert _native.lib().sc_version().startswith(b"semcode_hip")


def test_no_gpu_is_a_loud_error_not_a_fallback(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from semcode_amd import _native

    with pytest.raises(_native.ScError) as e:
        _native.Runtime(device=0)
    assert "no HIP device" in str(e.value)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under semcode_amd/ may reference it
    offenders = []
    for p in (ROOT / "semcode_amd").rglob("*"):
        if p.suffix in {".py", ".cpp", ".hip", ".h"} and "_obj" not in p.parts:
            t = p.read_text(errors="ignore")
            if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "libsc_oracle" in t or "sc_oracle.py" in t:
                offenders.append(str(p.relative_to(ROOT)))
    assert not offenders, offenders


def test_shared_runtime_fails_fast_without_a_device():
    """_native.shared_runtime must not hold the library lock while it creates the runtime (it once did: deadlock on first
    use).  Without a GPU the call has to come back with the library's own error, not hang."""
    import torch

    from semcode_amd import _native

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    with pytest.raises(_native.ScError, match="no HIP device"):
        _native.shared_runtime(0)
