"""Empirical roof of the sample arithmetic (k_roof) at several occupancies, with the
shader clock the chip holds meanwhile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_likelihood, TRUTH
like, flux = make_likelihood(0)
ctx = like._sync_device()
for wgs, thr in ((1, 256), (1, 512), (2, 512), (1, 1024), (2, 256), (4, 256), (3, 256)):
    ctx.set_option("roof_wgs_per_cu", wgs); ctx.set_option("roof_threads", thr)
    sec, slots, mhz = ctx.roof_probe(TRUTH, reps=100)
    print("%d WG/CU x %4d thr (%d waves/SIMD): %.3f ms, %.4g slots/s, clock %.0f MHz; cycles per chunk-wave per SIMD: %.1f" %
          (wgs, thr, wgs * thr // 256, sec * 1e3, slots / sec, mhz, sec * mhz * 1e6 * 1024 / (slots / 64)))
