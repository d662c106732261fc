"""GPU parity, randomised: a slice of tools/fuzz_scan.py (random alphabets, term lengths, document sizes around the
work-unit borders, folding, both position modes, the kernel's cross-check variants) against the CPU oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_scan.py")
    spec = importlib.util.spec_from_file_location("fuzz_scan", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [7, 8])
def test_scan_fuzz_slice(seed):
    done, err = _fuzz().run(iters=30, seed=seed, budget_s=60.0)
    assert err is None, err
    assert done >= 10
