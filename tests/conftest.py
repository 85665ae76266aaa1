import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def dev_lib():
    """The launches of the test go to libtulip_hip_dev.so (-DTULIP_DEV_VARIANTS=1): the kernel forms no step launches -- the
    recomputing C = 96 backward, the backward's split form, the profiled twins -- live only there."""
    from tulip_amd import _lib
    with _lib.dev_library():
        yield


def describe_flat_diff(eng, a, b, limit=8):
    """Where two flat parameter buffers differ: per parameter (or 'padding'), count and max |difference|."""
    a, b = a.detach().cpu(), b.detach().cpu()
    W = eng.params
    covered = a.new_zeros(a.numel(), dtype=bool)
    out = []
    for n in W.names:
        lo, hi = W.offset[n], W.offset[n] + W.numel[n]
        covered[lo:hi] = True
        d = (a[lo:hi] - b[lo:hi]).abs()
        k = int((d > 0).sum())
        if k:
            out.append(f"{n}: {k}/{hi - lo} differ, max {d.max().item():.3e}")
    d = (a - b).abs()[~covered]
    if int((d > 0).sum()):
        out.append(f"padding: {int((d > 0).sum())} differ, max {d.max().item():.3e}")
    return "; ".join(out[:limit]) + (f" ... (+{len(out) - limit} more)" if len(out) > limit else "")
