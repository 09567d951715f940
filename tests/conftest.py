import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

VARIANTS = [("thin_noalpha", True, True), ("thin_walpha", True, False),
            ("thick_noalpha", False, True), ("thick_walpha", False, False)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def g_sed():
    return np.load(os.path.join(GOLDEN, "sed.npz"))


@pytest.fixture(scope="session")
def g_lnl():
    return np.load(os.path.join(GOLDEN, "lnlike.npz"))


@pytest.fixture(scope="session")
def g_pb():
    return np.load(os.path.join(GOLDEN, "passbands.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def golden_bands(g_pb, names):
    """(wave, sedmult, normfac) per band from the reference-made tables."""
    return [(g_pb[str(n) + "/wave"], g_pb[str(n) + "/sedmult"], g_pb[str(n) + "/scalars"][0])
            for n in names]


def lnl_close(got, ref, rtol=1e-10):
    """|d lnL| <= 1e-10 max(1, |lnL|); -inf must match exactly."""
    got = np.asarray(got); ref = np.asarray(ref)
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    if not fin.any():
        return 0.0
    assert np.all(np.isfinite(got[fin]))
    err = np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))
    assert err.max() <= rtol, "max lnL error %g" % err.max()
    return err.max()
