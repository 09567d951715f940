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
def g_res():
    """Chain post-processing as the reference's own mbb_results computed it (make_golden_results.py)."""
    return np.load(os.path.join(GOLDEN, "results.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def golden_bands(g_pb, names):
    """(wave, sedmult, normfac) per band from the reference-made tables."""
    return [(g_pb[str(n) + "/wave"], g_pb[str(n) + "/sedmult"], g_pb[str(n) + "/scalars"][0])
            for n in names]


# ---- parity report: the largest error every tolerance check actually saw -----------------
# Written at the end of a session that made any check (the -m gpu run) to
# $MBB_PARITY_REPORT or gpurun_out/parity_report.json; the round's copy is committed
# under profiles/rNN/parity_report.json.
_PARITY = {}


def parity_record(kind, observed, bound):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    k = _PARITY.setdefault(test, {}).setdefault(kind, {"observed_max": 0.0, "bound": float(bound), "checks": 0})
    k["observed_max"] = max(k["observed_max"], float(observed))
    k["bound"] = min(k["bound"], float(bound))
    k["checks"] += 1


KERNEL_SOURCE_GLOBS = ("mbb_emcee_amd/csrc/*.hip", "mbb_emcee_amd/csrc/*.h", "mbb_emcee_amd/csrc/*.inc",
                       "mbb_emcee_amd/csrc/*.cpp", "include/*.h", "mbb_emcee_amd/build.py")


def kernel_sources_sha16():
    """Hash of everything the device library is built from: a parity report says which sources it
    was measured on (the GPU box has no .git), and tests/test_host_cpu.py holds the committed
    report to the sources of the tree it is committed in."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for pat in KERNEL_SOURCE_GLOBS:
        for f in sorted(glob.glob(os.path.join(ROOT, pat))):
            h.update(os.path.relpath(f, ROOT).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pytest_collection_modifyitems(config, items):
    """Tests that start other processes (tests/test_zz_gpu_processes.py) run LAST whatever order the files were named in
    on the command line: a harness of processes that fails under -x must not hide kernel tests behind it."""
    items.sort(key=lambda it: 1 if "test_zz_gpu_processes" in it.nodeid else 0)      # (stable: nothing else moves)


def pytest_deselected(items):
    if items:
        cfg = items[0].config
        cfg._mbb_deselected = getattr(cfg, "_mbb_deselected", 0) + len(items)


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    # the report is the GPU suite's (`-m gpu`); a CPU session checks the oracle against the golden
    # vectors with the same helpers and must not overwrite it
    if session.config.getoption("markexpr", "").strip() != "gpu" and "MBB_PARITY_REPORT" not in os.environ:
        return
    import json
    path = os.environ.get("MBB_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    worst = {}
    for t, kinds in _PARITY.items():
        for kind, v in kinds.items():
            w = worst.setdefault(kind, {"observed_max": 0.0, "test": None})
            if v["observed_max"] >= w["observed_max"]:
                w.update(observed_max=v["observed_max"], test=t, bound=v["bound"])
    head = None
    try:
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              timeout=10).stdout.decode().strip() or None
    except Exception:
        pass
    json.dump({"tolerances": "SURVEY.md 8(c): band flux / f_nu rtol 1e-12, xmerge atol 1e-10, "
                             "lnL |d| <= 1e-10 max(1, |lnL|), diagonal and covariance likelihoods alike",
               "exitstatus": int(exitstatus), "kernel_sources_sha16": kernel_sources_sha16(), "git_head": head,
               "tests_run": int(session.testscollected), "tests_deselected": int(getattr(session.config, "_mbb_deselected", 0)),
               "tests_failed": int(session.testsfailed), "tests_with_tolerance_checks": len(_PARITY),
               "worst_per_kind": worst, "tests": _PARITY},
              open(path, "w"), indent=1, sort_keys=True)


def rec_allclose(actual, desired, rtol=1e-7, atol=0.0, kind="value", **kw):
    """numpy.testing.assert_allclose that also records the largest error it saw, in units
    of its own tolerance's scale (relative where rtol is given, absolute otherwise)."""
    a = np.atleast_1d(np.asarray(actual, dtype=np.float64)); d = np.atleast_1d(np.asarray(desired, dtype=np.float64))
    a, d = np.broadcast_arrays(a, d)
    with np.errstate(invalid="ignore", divide="ignore"):
        fin = np.isfinite(a) & np.isfinite(d)
        if fin.any():
            if rtol > 0:
                err = np.abs(a - d)[fin] / np.maximum(np.abs(d)[fin], atol / rtol if atol > 0 else 1e-300)
                parity_record(kind + " (rel)", err.max(), rtol)
            else:
                parity_record(kind + " (abs)", np.abs(a - d)[fin].max(), atol)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol, **kw)


def lnl_close(got, ref, rtol=1e-10, kind="lnL"):
    """|d lnL| <= 1e-10 max(1, |lnL|); -inf must match exactly.
    `kind` labels the check in the parity report (e.g. "lnL (covariance)")."""
    got = np.asarray(got); ref = np.asarray(ref)
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    if not fin.any():
        return 0.0
    assert np.all(np.isfinite(got[fin]))
    err = np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))
    parity_record(kind + " / max(1,|lnL|)", err.max(), rtol)
    assert err.max() <= rtol, "max lnL error %g" % err.max()
    return err.max()
