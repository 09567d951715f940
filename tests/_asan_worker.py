"""Runs under LD_PRELOAD=libasan.so:libubsan.so (tests/test_host_cpu.py::test_sanitizers):
the host-only table builders of the product and the CPU oracle, both built with
-fsanitize=address,undefined (make -C oracle asan), driven through ctypes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ASAN = os.path.join(ROOT, "oracle", "_asan")
os.environ["MBB_ORACLE_LIB"] = os.path.join(ASAN, "libmbb_oracle_asan.so")


def main():
    import ctypes as C
    import _layout_checks as LC
    lib = LC.load(os.path.join(ASAN, "libmbb_hosttables_asan.so"))
    n = LC.run_all(lib, nrandom=int(os.environ.get("MBB_ASAN_LAYOUTS", "1000")))
    nb_, nc_, k_ = C.c_int(), C.c_int(), C.c_int()
    lib.mbbh_poly_counts(C.byref(nb_), C.byref(nc_), C.byref(k_))
    b = np.zeros(nb_.value * k_.value); c = np.zeros(nc_.value * k_.value)
    lib.mbbh_poly_tables(b.ctypes.data_as(LC._dp), c.ctypes.data_as(LC._dp))
    assert b[0] == 1.0 and c[0] == 0.0 and abs(c[1] - 0.125) < 1e-15      # b(0) = 1; C(y) = 1 - e^-y ~ y = t / 8 in row 0
    # the cross-process device registry (mbb_registry.cpp) under the sanitizers: a table of this worker's own
    os.environ["MBB_REGISTRY_NAME"] = "/mbb_hip_registry_asan_%d" % os.getpid()
    try:
        for fn in (lib.mbbh_registry_join, lib.mbbh_registry_leave):
            fn.argtypes = [C.c_uint32]
        lib.mbbh_registry_peers.argtypes = [C.c_uint32, C.c_int]
        for key in range(0x80000100, 0x80000100 + 20):          # (more keys than a slot holds: the rest go unseen, nothing overruns)
            lib.mbbh_registry_join(key)
        assert lib.mbbh_registry_peers(0x80000100, 1) == 0 and lib.mbbh_registry_peers(0x80000100 + 19, 0) == 0
        for key in range(0x80000100, 0x80000100 + 20):
            lib.mbbh_registry_leave(key); lib.mbbh_registry_leave(key)
    finally:
        try:
            os.unlink("/dev/shm" + os.environ.pop("MBB_REGISTRY_NAME"))
        except OSError:
            pass
    # the oracle on a slice of its golden vectors (the full comparison is test_oracle_golden.py)
    from oracle import oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "lnlike.npz"))
    gp = np.load(os.path.join(ROOT, "tests", "golden", "passbands.npz"))
    from conftest import golden_bands
    for cfg in ("cfg2", "cfg4"):
        k = cfg + "/thick_walpha"
        names = [str(x) for x in g[cfg + "/bands"]]
        cov = g[k + "/cov"] if cfg == "cfg4" else None
        orc = O.OracleLikelihood(g[k + "/flux"], g[k + "/unc"], bands=golden_bands(gp, names), cov=cov,
                                 has_uplim=[0, 1, 1, 1, 0, 0], uplim=g[k + "/uplim"])
        got = orc(g[k + "/pars"][:60], nthreads=1)
        ref = g[k + "/lnl"][:60]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isneginf(got), np.isneginf(ref))
        assert np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1, np.abs(ref[fin]))) < 1e-12
    s = O.OracleSED(20.0, 1.9, 250.0, 3.5, 50.0)
    assert abs(s.wavemerge - 109.5506829) < 1e-3 and s.max_wave() > 0
    print("ASAN_OK layouts=%d" % n)


if __name__ == "__main__":
    main()
