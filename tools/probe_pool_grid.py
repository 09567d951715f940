#!/usr/bin/env python3
"""The pool pattern (several processes in a loop of boundary calls on one GPU) with the share rule off ("serve" 2) and servers of
a given width ("serve_grid"): two workers on half the CUs each, both on every CU (round 4's arrangement), three workers on a
third each (their 125 rows do not fit: launches), one alone.      python tools/probe_pool_grid.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
from tools import probe_pool as P
for world, grid in ((2, 128), (2, 0), (3, 85), (1, 128), (1, 0)):
    print("==== world %d, serve 2 (regardless of peers), serve_grid %d" % (world, grid), flush=True)
    P.run(world, 3000, 2, {"MBB_POOL_OPTIONS": json.dumps({"serve_grid": grid, "serve_lease_us": 0}), "MBB_POOL_HAS_PEERS_INFO": "1"})
