"""
pytest configuration.

Markers
  gpu              needs an MI355X (run by the driver with `-m gpu` on the GPU box).  These are the
                   parity tests proper: they call the HIP path through the C ABI and compare with
                   the CPU oracle / committed golden fixtures.  They never read /root/reference.
  (default)        CPU-only: oracle vs golden vectors, host logic, C-ABI load/export checks,
                   world_size-2 gloo tests of the sharded path.

The oracle/ package is test infrastructure: only tests import it (as the checker).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    config.addinivalue_line("markers", "needs_reference: imports /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import ref_import
    have_ref = ref_import.available()
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present on this box"))
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def s1_inputs():
    """S1 (BASELINE.json configs[0]): seed-0 Gaussian, q [1000,768] then d [100000,768], one generator."""
    import torch
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1000, 768, generator=g)
    d = torch.randn(100000, 768, generator=g)
    gold = np.load(os.path.join(GOLDEN, "config1.npz"))
    # guard: the generator must reproduce the data the golden outputs were computed on
    assert abs(float(q.double().sum()) - float(gold["checksum_q"])) < 1e-6
    assert abs(float(d.double().sum()) - float(gold["checksum_d"])) < 1e-6
    return q, d, gold


def gap_tolerance(q, x):
    """Near-tie width for fp32 dot products of different summation order: 4 * 2^-24 * |q| * |x| (SURVEY H-1)."""
    qn = float(np.linalg.norm(np.asarray(q, np.float64), axis=1).max())
    xn = float(np.linalg.norm(np.asarray(x, np.float64), axis=1).max())
    return 4.0 * 2.0 ** -24 * qn * xn
