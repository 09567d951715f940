"""MCMC steps/s of the host-driven stretch-move sampler (one likelihood launch per
half-step across the host boundary) next to the device-resident one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
p0 = walkers(1)[:250]
s = mbb.EnsembleSampler(250, 5, like, seed=1)
pos, lnp, _ = s.run_mcmc(p0, 50)
t0 = time.perf_counter(); pos, lnp, _ = s.run_mcmc(pos, 500); dt = time.perf_counter() - t0
print("host sampler   : %.0f steps/s (%.1f us/step), acceptance %.2f" % (500 / dt, dt / 500 * 1e6, np.mean(s.acceptance_fraction)))
d = mbb.DeviceEnsembleSampler(250, 5, like, seed=1)
d.run_mcmc(p0, 50)
t0 = time.perf_counter(); d.run_mcmc(None, 500); dt = time.perf_counter() - t0
print("device sampler : %.0f steps/s (%.1f us/step), acceptance %.2f" % (500 / dt, dt / 500 * 1e6, np.mean(d.acceptance_fraction)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); s.run_mcmc(pos, 300); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
