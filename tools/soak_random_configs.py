#!/usr/bin/env python3
"""tests/test_gpu_parity.py::test_random_sampler_configurations_all_forms_equal over many more seeds than the suite runs (10):
random bands, model variant, priors, limits, covariance, ensemble size, run lengths, workgroup width -- every sampler form
against the plain launch train, bit for bit.      python tools/soak_random_configs.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mbb_emcee_amd as mbb
import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_random_sampler_configurations_all_forms_equal(mbb, seed)
    except AssertionError as e:
        bad.append(seed)
        print("seed %d: %s" % (seed, str(e)[:300]), flush=True)
    if (seed - first) % 10 == 9:
        print("... %d seeds, %d wrong, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
print("random configurations: %d of %d seeds wrong %s" % (len(bad), count, bad))
sys.exit(1 if bad else 0)
