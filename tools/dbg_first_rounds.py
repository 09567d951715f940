import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
smp = mbb.DeviceEnsembleSampler(250, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:250], 60, storechain=False)
def A(): smp.advance_async(5); ctx.sync(); return smp.advance_timed(20)
def B(): smp.advance_timed(5); return smp.advance_timed(20)
def C(): return smp.advance_timed(20)
def D(): smp.advance_async(20); ctx.sync(); return smp.advance_timed(20)
def E(): smp.advance_async(5); ctx.sync(); smp.advance_async(5); ctx.sync(); return smp.advance_timed(20)
def F(): smp.advance_async(6); ctx.sync(); return smp.advance_timed(20)
def G(): smp.advance_async(7); ctx.sync(); return smp.advance_timed(20)
for rep in range(2):
    for name, fn in (("A async5,timed20", A), ("B timed5,timed20", B), ("C timed20", C), ("D async20,timed20", D), ("E async5,async5,timed20", E), ("F async6,timed20", F), ("G async7,timed20", G)):
        out = np.array([fn() for _ in range(40)])
        print("%-26s wall %.1f stream %.1f" % (name, np.median(out[:, 0]) * 1e6, np.median(out[:, 1]) * 1e3), flush=True)
