import sys; sys.path.insert(0,'/root/repo')
import numpy as np, mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
def t(label, debug=0):
    s = mbb.DeviceEnsembleSampler(250, 5, like, seed=7)
    s.run_mcmc(walkers(1)[:250], 20, storechain=False)
    ctx.set_option("debug", debug)
    s.advance_async(200); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); s.advance_async(1000); ctx.record(e1); ctx.sync()
    print("%-40s %.3f us per step" % (label, ctx.elapsed_ms(e0, e1)))
    ctx.set_option("debug", 0)
ctx.set_option("persistent_sampler", 0); t("one launch per half-step")
ctx.set_option("persistent_sampler", 1); t("one launch per run")
# (with the wait and the arrival counting compiled out of the one-launch form the same run took
#  15.17 us per step, and 15.66 with the counting only: the body of a half-step is 7.6 us in
#  either form -- round-2 diagnostic builds, profiles/r02/persistent_sampler.txt)
