#!/usr/bin/env python3
"""Form 9 (k_flowa) against the launch train on a small ensemble with a short poll budget: a protocol fault shows
as a give-up (fallback count), not as a hang.   python tools/dbg_form9.py [walkers] [W] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from mbb_emcee_amd import _native
if os.environ.get("MBB_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["MBB_LIB"])        # (another build of the library)
from bench import make_likelihood, TRUTH
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 26
W = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
like, flux = make_likelihood(0)
ctx = like._sync_device()
ctx.set_option("flow_spin_log2", 12)
p0 = TRUTH * (1.0 + 0.02 * np.random.RandomState(1).normal(size=(nw, 5)))
out = []
for look in (0, 1):
    ctx.set_option("lookahead_sampler", look); ctx.set_option("resident_sampler", 2); ctx.set_option("resident_ahead", 2)
    ctx.set_option("resident_walkers", W)
    s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=3)
    t0 = time.time()
    import warnings
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        r = s.run_mcmc(p0, steps)
    print("look", look, "form", ctx.info("last_kernel_form"), "fallbacks", ctx.info("flow_fallbacks"), "%.2f s" % (time.time() - t0),
          [str(w.message)[:60] for w in wlist], flush=True)
    out.append((r[0], r[1], s.chain.copy()))
print("equal:", [bool(np.array_equal(x, y)) for x, y in zip(*out)])
