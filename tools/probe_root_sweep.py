"""Wide random sweep of the thick merge-point solve: fp64 Newton evaluations taken
(option `debug`) and the constructor scalars against the CPU oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from oracle.oracle import OracleSED
from bench import BANDS, TRUTH

rng = np.random.RandomState(7)
n = 200000
p = np.column_stack([np.exp(rng.uniform(np.log(2), np.log(300), n)),      # T
                     rng.uniform(0.0, 5.0, n),                             # beta
                     np.exp(rng.uniform(np.log(1), np.log(5000), n)),      # lambda0
                     np.exp(rng.uniform(np.log(0.05), np.log(12), n)),     # alpha
                     np.exp(rng.uniform(np.log(0.1), np.log(1000), n))])   # fnorm
like = mbb.likelihood(response=True)
like.set_phot(BANDS, np.ones(8), np.ones(8))
like.set_lowlim("T", 0.0)
ctx = like._sync_device()
d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
ctx.set_option("debug", 1)
ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 1); ctx.sync()
st = d_st.download(np.int32, n)
ctx.set_option("debug", 0)
print("row status:", np.bincount(st & 255), " fp64 evaluations:", np.bincount(st >> 8))
out, st2 = mbb._native.default_context().sed_prologue(p, False, False, 500.0, want_peak=True)
m = 20000
ref = np.full((m, 3), np.nan)
for i in range(m):
    try:
        o = OracleSED(*p[i], wavenorm=500.0)
        ref[i] = (o.s.normfac, o.s.xmerge, o.s.kappa)
    except ValueError:
        pass
for k, name in enumerate(["normfac", "xmerge", "kappa"]):
    r = ref[:, k]; g = out[:m, k]
    ok = np.isfinite(r) & np.isfinite(g) & (r != 0)
    err = np.abs(g[ok] / r[ok] - 1)
    print(name, "max rel err vs oracle: %.3e, 99.9th percentile %.3e" % (err.max(), np.percentile(err, 99.9)),
          "(%d rows)" % ok.sum())
    if name == "xmerge":
        i = np.flatnonzero(ok)[np.argmax(err)]
        print("  worst row:", p[i], "gpu", g[i], "oracle", r[i])
