"""Two (or more) processes in a loop of boundary calls on one GPU (emcee's pool pattern, mbb_fit.py:80-81): reports of
tests/_pool_worker.py, for serve on / off.  python tools/probe_pool.py [world] [ncalls]"""
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(world, ncalls, serve, extra=None):
    import mbb_emcee_amd as mbb
    g = np.load(os.path.join(ROOT, "tests", "golden", "lnlike.npz"))
    like = mbb.likelihood(response=True)
    like.set_phot([str(b) for b in g["cfg2/bands"]], g["cfg2/thick_walpha/flux"], g["cfg2/thick_walpha/unc"])
    pars = np.ascontiguousarray(np.tile(g["cfg2/thick_walpha/pars"], (2, 1))[:125])
    ctx = like._sync_device()
    ctx.set_option("serve", 0)
    want = like(pars).copy()
    d = tempfile.mkdtemp()
    pickle.dump(like, open(os.path.join(d, "like.pkl"), "wb"))
    np.save(os.path.join(d, "pars.npy"), pars); np.save(os.path.join(d, "want.npy"), want)
    del like, ctx
    env = dict(os.environ, OMP_NUM_THREADS="2", MBB_POOL_SERVE=str(serve))
    env.update(extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_pool_worker.py"), d, str(r), str(world), str(ncalls)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for pr in procs:
        try:
            out = pr.communicate(timeout=200)[0].decode(errors="replace")
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            print("TIMEOUT"); return
        print(out.strip()[-500:])
    for r in range(world):
        rep = json.load(open(os.path.join(d, "out.%d.json" % r)))
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rep.items()}))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    for serve in (0, 1):
        print("---- world %d, %d calls of 125 rows each, serve %d" % (world, ncalls, serve))
        run(world, ncalls, serve, {"MBB_POOL_HAS_PEERS_INFO": "1"} if os.environ.get("MBB_POOL_HAS_PEERS_INFO") else None)
    print("---- one process alone, serve 1")
    run(1, ncalls, 1)
