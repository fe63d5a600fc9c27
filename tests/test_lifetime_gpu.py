"""GPU: handle lifetimes.  The runtime is reference counted in C (sc_runtime_retain / _release): indexes and encoders keep
it alive, so handles may be destroyed in any order -- by the caller, by the garbage collector, or at interpreter exit.

Background: profiles/r1l_ivf_1Mx768.log records `timeout: the monitored command dumped core` after
scripts/bench_ivf.py had printed its result.  At that time Runtime.__del__ / Index.__del__ destroyed their handles during
interpreter shutdown in whatever order the module globals were cleared; sc_index_destroy then called hipSetDevice /
hipStreamSynchronize through ix->rt, which sc_runtime_destroy had already freed."""
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

from oracle import sc_oracle as orc
from semcode_amd import _native

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_children_outlive_a_closed_runtime():
    rt2 = _native.Runtime(device=0)
    ix = _native.Index(rt2, 64, metric="L2")
    enc = _native.Encoder(rt2, dict(vocab=100, hidden=128, layers=1, heads=2, ffn=256, max_pos=64), synth_seed=1)
    X = orc.synth(3000, 64, seed=1)
    ix.add(X)
    rt2.close()  # the creator's reference goes; the stream and the struct stay until the last child is destroyed
    with pytest.raises(RuntimeError, match="runtime is closed"):
        _native.Index(rt2, 64)
    Q = orc.synth(3, 64, seed=2)
    d, r = ix.search(Q, k=5)
    od, orow = orc.search(X, Q, 5, "L2")
    assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
    out = enc.embed_ids(np.ones((2, 32), np.int32), np.array([32, 5], np.int32))
    assert out.shape == (2, 128) and np.isfinite(out).all()
    ix.close()
    enc.close()  # last reference: the runtime is torn down here


def test_teardown_after_fallback_release_scratch_and_train():
    """The sequence of the run that dumped core: batched search with uncertified queries (exact fallback buffers), scratch
    release, IVF training (quantizer sub-index, second corpus copy), probes, then close -- index first or runtime first."""
    for runtime_first in (False, True):
        rt2 = _native.Runtime(device=0)
        ix = _native.Index(rt2, 96, metric="L2", kind="IVF_FLAT", nlist=16)
        base = orc.synth(20_000, 96, seed=3)
        base[5000:5600] = base[5000] + 1e-4 * orc.synth(600, 96, seed=4)  # 600 near-duplicates: the certificate cannot separate them
        ix.add(base)
        Q = np.concatenate([base[5000:5001] + 1e-5, orc.synth(39, 96, seed=5)])
        ix.set_search_mode("batched")
        _native.diag_set_option("collect_pass", 0)  # (the collect pass would answer these queries: the sequence needs the exact fallback)
        try:
            d, r = ix.search(Q, k=10)
        finally:
            _native.diag_set_option("collect_pass", 1)
        st = ix.last_search_stats()
        assert st["path"] == "batched" and st["uncertified"] >= 1
        od, orow = orc.search(base, Q, 10, "L2")
        assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
        ix.set_search_mode("auto")
        ix.release_scratch()
        ix.train(niter=3)
        ix.search(Q[:2], k=10, nprobe=4)
        ix.release_scratch()
        ix.train(niter=2)  # retrain: the old quantizer is destroyed and replaced
        ix.search(Q, k=10, nprobe=4)
        if runtime_first:
            rt2.close()
            ix.close()
        else:
            ix.close()
            rt2.close()


def test_interpreter_exit_with_live_handles_is_clean():
    """A script that never closes anything and lets interpreter shutdown collect its handles must exit with status 0."""
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        from semcode_amd import _native
        rt = _native.Runtime(0)
        ix = _native.Index(rt, 64, metric="IP", kind="IVF_FLAT", nlist=8)
        ix.fill_synthetic(5000, seed=1)
        ix.train(niter=2)
        other = _native.Index(rt, 64)
        d, r = ix.search(np.ones((20, 64), np.float32), k=5, nprobe=2)
        del rt  # the globals go in arbitrary order from here on
        print("done", int(r[0, 0] >= 0), flush=True)
    """ % str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "done 1" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
