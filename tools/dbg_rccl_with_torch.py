#!/usr/bin/env python3
"""The launcher's situation at N > 1: torch (with its bundled librccl / libamdhip64 / libhsa-runtime64) is in the process
BEFORE libmbb_hip.so; then a one-rank RCCL communicator through the C-ABI, an all-gather, the sharded boundary call,
tear-down and a clean exit.   python tools/dbg_rccl_with_torch.py"""
import os, sys
import torch                                    # noqa: F401  (first, as in bench.py's ranks)
import torch.distributed                        # noqa: F401
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers
from mbb_emcee_amd import parallel
like, flux = make_likelihood(0)
ctx = like._sync_device()
uid = ctx.comm_unique_id()
ctx.comm_init(1, 0, uid)
p = np.ascontiguousarray(walkers(1)[:125])
full, st = ctx.lnlike_allgather(p, 1)
assert np.array_equal(full, like(p))
ctx.comm_destroy()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.split("\n") if any(k in l for k in ("librccl", "libamdhip64", "libhsa-runtime"))})
print("ok; ROCm libraries in the process:", libs)
