"""Worker of test_served_boundary_in_two_processes_on_one_gpu: emcee's pool pattern (reference mbb_fit.py:80-81 with
threads > 1: the likelihood object pickled into worker processes).  Each worker unpickles the SAME likelihood, waits
until every worker is ready, and makes `ncalls` boundary calls of 125 rows in a loop -- what a pool worker does for
its share of every half-step -- checking every result bit for bit and timing every call.

    python tests/_pool_worker.py <dir> <rank> <world> <ncalls>

<dir> holds like.pkl, pars.npy, want.npy; the worker writes <dir>/out.<rank>.json."""
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    d, rank, world, ncalls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import mbb_emcee_amd  # noqa: F401
    like = pickle.load(open(os.path.join(d, "like.pkl"), "rb"))
    pars = np.load(os.path.join(d, "pars.npy"))
    want = np.load(os.path.join(d, "want.npy"))
    ctx = like._sync_device()
    for opt, val in json.loads(os.environ.get("MBB_POOL_OPTIONS", "{}")).items():
        ctx.set_option(opt, int(val))
    ctx.set_option("serve", 0)
    first = like(pars)                                   # (context, tables, the first launch: outside the loop)
    ctx.set_option("serve", int(os.environ.get("MBB_POOL_SERVE", "1")))
    ok = bool(np.array_equal(first, want, equal_nan=True))
    open(os.path.join(d, "ready.%d" % rank), "w").close()
    t_wait = time.time()
    while not all(os.path.exists(os.path.join(d, "ready.%d" % r)) for r in range(world)):
        if time.time() - t_wait > 120:
            raise SystemExit("the other workers never became ready")
        time.sleep(0.0005)
    peers_at_start = ctx.info("device_peers") if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None
    ts = np.empty(ncalls)
    bad = 0
    t0 = time.perf_counter()
    for i in range(ncalls):
        a = time.perf_counter()
        got = like(pars)
        ts[i] = time.perf_counter() - a
        if not np.array_equal(got, want, equal_nan=True):
            bad += 1
    wall = time.perf_counter() - t0
    out = {"rank": rank, "first_ok": ok, "bad_calls": bad, "calls": ncalls, "wall_s": wall,
           "t_start": t0, "t_end": t0 + wall,
           "p50_us": float(np.median(ts) * 1e6), "p99_us": float(np.percentile(ts, 99) * 1e6), "max_us": float(ts.max() * 1e6),
           "calls_over_1ms": int((ts > 1e-3).sum()), "calls_over_100us": int((ts > 1e-4).sum()),
           "serve_requests": ctx.info("serve_requests"), "serve_fallbacks": ctx.info("serve_fallbacks"),
           "serve_enabled_at_end": ctx.info("serve_enabled"), "serve_rests": ctx.info("serve_rests"), "serving_at_end": ctx.info("serving"),
           "serve_peer_yields": ctx.info("serve_peer_yields") if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None,
           "serve_grid_at_end": ctx.info("serve_grid") if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None,
           "serve_resizes": ctx.info("serve_resizes") if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None,
           "busy_at_end": ctx.info("device_busy") if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None,
           "peers_at_start": peers_at_start}
    json.dump(out, open(os.path.join(d, "out.%d.json" % rank), "w"))
    print("POOL_OK %d" % rank if ok and bad == 0 else "POOL_BAD %d" % rank, flush=True)


if __name__ == "__main__":
    main()
