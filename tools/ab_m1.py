#!/usr/bin/env python3
"""Interleaved A/B of M1 (likelihood.__call__ on host arrays, by the host's clock) for two builds of the library in ONE process:
    python tools/ab_m1.py <libA.so> <libB.so> [rounds]
Process-to-process the same build differs by 10-20 % (where the host thread runs); inside one process the builds alternate."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_ALL = dict(_native.SIGNATURES)


def side(libpath):
    import ctypes
    _native._lib = None
    _native.LIB_PATH = os.path.abspath(libpath)
    probe = ctypes.CDLL(_native.LIB_PATH)
    _native.SIGNATURES.clear()
    _native.SIGNATURES.update({k: v for k, v in _ALL.items() if hasattr(probe, k)})
    from bench import make_likelihood
    like, _ = make_likelihood(0)
    return like, like._sync_device()


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    from bench import walkers
    sides = {"A": side(a), "B": side(b)}
    res = {}
    want = {}
    for rnd in range(rounds):
        for name, (like, ctx) in sides.items():
            for n in (125, 1):
                p = np.ascontiguousarray(walkers(1)[:n])
                arg = p if n > 1 else p[0].copy()
                for mode in ("served", "launch"):
                    ctx.set_option("serve", 1 if mode == "served" else 0)
                    for _ in range(30):
                        got = like(arg)
                    w = want.setdefault(n, np.array(got))
                    assert np.array_equal(np.array(got), w), (name, n, mode)
                    ts = np.empty(1000)
                    for i in range(1000):
                        t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
                    res.setdefault((name, n, mode), []).append(np.median(ts) * 1e6)
                    ctx.set_option("serve", 0)          # (the server leaves: the other build's turn)
    print("A = %s\nB = %s" % (a, b))
    for n in (125, 1):
        for mode in ("served", "launch"):
            va, vb = np.array(res[("A", n, mode)]), np.array(res[("B", n, mode)])
            print("rows %3d %-7s  A %6.2f us   B %6.2f us   B/A %.3f   (rounds A %s | B %s)" % (
                n, mode, np.median(va), np.median(vb), np.median(vb) / np.median(va),
                " ".join("%.2f" % x for x in va), " ".join("%.2f" % x for x in vb)))


if __name__ == "__main__":
    main()
