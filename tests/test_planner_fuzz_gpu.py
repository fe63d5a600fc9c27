"""Every path the search planner can take, on random index shapes, against the exact path of the same index -- bit for bit.

The sweep itself is `scripts/planner_fuzz.py` (400 cases, each searched again after overwrites / appends: `profiles/r3z_planner_fuzz.log`); here 40 cases of one fixed seed (~10 s).
The exact paths it compares with are the ones `tests/test_scan_gpu.py` / `tests/test_ivf_gpu.py` hold against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_planner_paths_agree_with_the_exact_path():
    from scripts import planner_fuzz

    assert planner_fuzz.sweep(seed=3, cases=40) == 0
