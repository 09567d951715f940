#!/usr/bin/env python3
"""Interleaved A/B of the device sampler's step (form 7, 250 walkers of cfg2; HIP events, 1000 steps) for several builds of the
library in one process:   python tools/ab_sampler_step.py <lib.so> [<lib.so> ...] [rounds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
import mbb_emcee_amd as mbb
_ALL = dict(_native.SIGNATURES)


def side(libpath):
    import ctypes
    _native._lib = None
    _native.LIB_PATH = os.path.abspath(libpath)
    probe = ctypes.CDLL(_native.LIB_PATH)
    _native.SIGNATURES.clear()
    _native.SIGNATURES.update({k: v for k, v in _ALL.items() if hasattr(probe, k)})
    from bench import make_likelihood, walkers, NW_PER_GPU
    like, _ = make_likelihood(0)
    ctx = like._sync_device()
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=7)
    pos, lnp, _ = smp.run_mcmc(walkers(1)[:NW_PER_GPU], 20, storechain=False)
    return like, ctx, smp, (pos.copy(), lnp.copy())


libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 5
sides = [side(l) for l in libs]
assert all(np.array_equal(s[3][0], sides[0][3][0]) and np.array_equal(s[3][1], sides[0][3][1]) for s in sides), "the builds' chains differ"
res = [[] for _ in libs]
for r in range(rounds):
    for i, (like, ctx, smp, _) in enumerate(sides):
        smp.advance_async(100); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(1000); ctx.record(e1); ctx.sync()
        res[i].append(ctx.elapsed_ms(e0, e1))
base = np.median(res[0])
for l, v in zip(libs, res):
    print("%-40s %.3f us per step   x%.4f   (%s)" % (os.path.basename(l), np.median(v), np.median(v) / base, " ".join("%.3f" % x for x in v)))
