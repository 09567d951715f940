"""The GPU tests that START OTHER PROCESSES (ranks on this one GPU, pool workers, `bench.py --gpus 2`), in a file of
their own that pytest collects LAST (files in name order): whatever goes wrong with a harness of processes can hide
only other tests of this kind behind `-x`, never a kernel test.  Every group of ranks is started by the test itself and
meets through files (tests/_filecomm.py)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mbb():
    import mbb_emcee_amd
    return mbb_emcee_amd


def _cfg2_like(mbb, g_lnl):
    like = mbb.likelihood(response=True)
    like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl["cfg2/thick_walpha/flux"],
                  g_lnl["cfg2/thick_walpha/unc"])
    return like


def test_bench_with_two_ranks_sharing_this_gpu():
    """`python bench.py --gpus 2` as the driver starts it, end to end, with the two ranks as processes on this one GPU
    (--oversubscribe: a rehearsal, said so on the line): the supervisor, the gloo side channel, the one-hop exchange
    through hipIpc mappings set up / rehearsed against the unsharded sampler / preconditioned / timed / validated, one
    JSON line with a value and the ranks' copies of the ensemble agreeing.  (RCCL needs a device per rank and the
    one-launch exchange every workgroup of both ranks resident: neither is possible here, and the line says so.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    side = os.path.join(tempfile.mkdtemp(), "bench_full.json")
    env = dict(os.environ, MBB_BENCH_WALKERS_PER_GPU="32", MBB_BENCH_FULL=side)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                          "--oversubscribe"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] and d["value"] > 0 and d["ranks_agree"] is True, d
    assert d["valid_for_scaling"] is False and "rehearsal" in d["config"]["note"]
    assert "2000 untimed steps" in d["config"]["preconditioning"] and "rehearsal" in d["config"]
    # the line is the short one (below 4 KB: one word per exchange), the detail is in the side file
    assert len(lines[0]) < 4096
    ev = d["exchange_validation"]
    assert ev["ipc-launches"] == "ok" and ev["rccl"].startswith("skipped") and ev["ipc"].startswith("skipped"), ev
    full = json.load(open(side))
    ev = full["exchange_validation"]
    assert ev["ipc-launches"]["ok"] is True and ev["rccl"]["ok"] is None and ev["ipc"]["ok"] is None, ev
    assert ev["ipc-launches"]["us_per_step"] > 0 and full["value"] == pytest.approx(d["value"], rel=1e-5)


def _run_pool_workers(mbb, g_lnl, world, ncalls, env_extra=None):
    """Start `world` processes that each unpickle the same likelihood and make `ncalls` boundary calls of 125 rows at
    the same time on this one GPU (tests/_pool_worker.py); returns their reports."""
    import json, pickle, subprocess, sys, tempfile
    from conftest import ROOT
    like = _cfg2_like(mbb, g_lnl)
    pars = np.ascontiguousarray(np.tile(g_lnl["cfg2/thick_walpha/pars"], (2, 1))[:125])
    ctx = like._sync_device()
    ctx.set_option("serve", 0)
    want = like(pars).copy()
    ctx.set_option("serve", 1)
    d = tempfile.mkdtemp()
    pickle.dump(like, open(os.path.join(d, "like.pkl"), "wb"))
    np.save(os.path.join(d, "pars.npy"), pars); np.save(os.path.join(d, "want.npy"), want)
    del like
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_pool_worker.py"), d, str(r), str(world), str(ncalls)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    texts = []
    for pr in procs:
        try:
            texts.append(pr.communicate(timeout=240)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                              # exactly the processes started above
            raise AssertionError("a pool worker did not finish within 240 s")
    for r, (pr, t) in enumerate(zip(procs, texts)):
        assert pr.returncode == 0 and "POOL_OK %d" % r in t, t[-3000:]
    return [json.load(open(os.path.join(d, "out.%d.json" % r))) for r in range(world)]


def test_served_boundary_in_two_processes_on_one_gpu(mbb, g_lnl):
    """emcee's pool (reference mbb_fit.py:80-81 with threads > 1: the likelihood pickled into worker processes): two
    processes, each with its own copy of the same likelihood, each in a loop of boundary calls -- on ONE GPU.  A
    resident server (k_serve) holds a CU per workgroup with most of its LDS: nothing of another process fits on those.
    Round 4's server sat on every CU and one worker's call waited 42 ms for the other's whole loop.  Now a server is as wide
    as the calls have rows and no wider than the process's share of the device (the CUs divided by the processes that are
    making boundary calls on it: a registry in shared memory, csrc/mbb_registry.cpp), so two workers of 125 rows each have
    their servers side by side; three have a share of 80 CUs each and their servers' workgroups take two rows of a request.
    Here: every result of every process bit for bit right, the loops done in a bounded time, nobody waiting long for another
    (no call beyond 20 ms), everybody served (the process of this test session holds contexts on the device as well, but
    makes no calls meanwhile: it is not in the way)."""
    import json
    from conftest import ROOT
    ncalls = 3000
    rep = _run_pool_workers(mbb, g_lnl, 2, ncalls, {"MBB_POOL_HAS_PEERS_INFO": "1"})
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "pool_two_processes.json"), "w"), indent=1)
    except OSError:
        pass
    overlap = min(r["t_end"] for r in rep) - max(r["t_start"] for r in rep)
    for r in rep:
        assert r["first_ok"] and r["bad_calls"] == 0 and r["calls"] == ncalls, r
        assert r["wall_s"] < 5.0 and r["max_us"] < 20000.0, r          # nobody starved: 3000 calls take ~0.03-0.1 s
        assert r["serve_enabled_at_end"] == 1 and r["serve_rests"] == 0, r   # the feature did not strike out either
        assert r["peers_at_start"] >= 1, r                             # (each sees the other, and this session's process)
    assert overlap > 0.0, rep                                          # (the two loops did run at the same time)
    # side by side: all but the first few calls of each were answered by its own resident kernel, 128 workgroups wide
    assert all(r["serve_requests"] >= 0.9 * ncalls and r["serve_fallbacks"] <= 3 for r in rep), rep
    assert all(r.get("serve_grid_at_end", 128) == 128 for r in rep), rep
    # three workers of 125 rows: 375 CUs are not there -- a share of 80 each (10 on each of the 8 XCDs: the dispatcher deals a
    # kernel's workgroups to the XCDs in turn, three times 11 do not fit into an XCD's 32 CUs), two rows a workgroup
    rep3 = _run_pool_workers(mbb, g_lnl, 3, 1500, {"MBB_POOL_HAS_PEERS_INFO": "1"})
    for r in rep3:
        assert r["first_ok"] and r["bad_calls"] == 0 and r["calls"] == 1500 and r["max_us"] < 20000.0, r
        assert r["serving_at_end"] == 1 and r["serve_grid_at_end"] == 80 and r["serve_fallbacks"] <= 3, r
        assert r["serve_requests"] >= 0.9 * 1500, r


def _run_ranks(script, nproc, marker, extra_env=None):
    """`nproc` ranks of tests/<script> as children of this process, met through files (tests/_filecomm.py): no port is
    ever picked, no launcher in between.  Round 5's driver run lost this test -- and, with -x, the 27 behind it -- to a
    port that was free when the test looked and taken when torch.distributed.run went to listen on it."""
    from conftest import ROOT
    from _filecomm import launch
    ok, text = launch(os.path.join(ROOT, "tests", script), nproc, extra_env=extra_env, timeout=300.0)
    if not ok or marker not in text:
        try:                                           # the whole transcript: the first failing rank's lines are early in it
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "%s_%d_ranks.txt" % (script, nproc)), "w").write(text)
        except OSError:
            pass
    assert ok and marker in text, text[-6000:]
    return text


def _run_xchg_worker(nproc, extra_env=None):
    _run_ranks("_xchg_worker.py", nproc, "XCHG_OK", extra_env)


def test_one_hop_exchange_two_processes_one_gpu():
    """The sharded device sampler with the one-hop peer-write exchange (mbb_xchg_*): two
    processes on this one GPU map each other's copy of the ensemble through hipIpc, each
    moves its block of every half-ensemble and stores what it decides into both copies -- as
    one launch per run handing over row by row across the ranks (k_lnlike SMODE 6) and as one
    launch per half-step with the moved rows exchanged after it (SMODE 2); chains, final state
    and counts are bitwise those of the unsharded run (tests/_xchg_worker.py).  On a multi-GPU
    node the same protocol crosses xGMI instead of staying inside one device; that part cannot
    be exercised on a one-GPU box."""
    _run_xchg_worker(2)


def test_one_hop_exchange_three_processes_one_gpu():
    """The same with three ranks (72 walkers, 12 movers per rank and half-step): every rank's
    movers wait for decisions made on two other ranks, every rank's lag guard for two others'
    progress words.  (Four and five ranks were run by hand -- profiles/r02/
    sharded_one_launch_rehearsal.txt --; the suite stays clear of the box's limit of six processes
    on the GPU, which the test runner and the launcher count towards.)"""
    _run_xchg_worker(3, {"MBB_XCHG_TEST_WALKERS": "72", "MBB_XCHG_TEST_SKIP_LOST_PEER": "1"})


def _standin_rccl():
    """tests/rccl_standin/librccl_standin.so, built here when it is not (hipcc is on the GPU box: the same image)."""
    import subprocess
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "rccl_standin")
    so, src = os.path.join(d, "librccl_standin.so"), os.path.join(d, "rccl_standin.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lrt"])
    return so


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_paths_with_several_ranks_on_one_gpu(world):
    """Everything the library does with a communicator of MORE THAN ONE rank -- mbb_comm_init, mbb_allgather_f64 in and out
    of place, mbb_lnlike_allgather (rank-major offsets, the ragged last block, rows through the BAR or copied),
    mbb_lnlike_allgather_device with its in-place slice, the device sampler's gathers of moved rows, chain and counts --
    executed with 2 and 3 ranks as processes on this GPU and held bitwise to the unsharded evaluation
    (tests/_rccl_worker.py).  RCCL itself needs a device per rank, so the ranks bind a stand-in for its four entry
    points through MBB_RCCL_LIB (tests/rccl_standin/: an all-gather through hipIpc-mapped staging buffers).  What this
    cannot show is RCCL's own behaviour over xGMI; what it does show is that the code around the collective -- never run
    with nranks > 1 before round 6 -- is right.  Role replaced: emcee's pool, reference mbb_fit.py:80-81."""
    text = _run_ranks("_rccl_worker.py", world, "RCCL_RANKS_OK 0", {"MBB_RCCL_LIB": _standin_rccl()})
    for r in range(world):
        assert "RCCL_RANKS_OK %d" % r in text


def test_bench_rccl_legs_with_two_ranks_sharing_this_gpu():
    """bench.py's two RCCL legs -- the `rccl` exchange of the device sampler (one launch per half-step + an in-place
    all-gather of the moved rows) and `boundary_sharded`, north_star's own split (likelihood.__call__ sharded, ONE
    all-gather of lnprob per call) -- had never run with more than one rank: real RCCL wants a device per rank.  Here
    `python bench.py --gpus 2 --exchange rccl` end to end with the two ranks on this GPU and the collective library named
    by MBB_RCCL_LIB (tests/rccl_standin/): the unique id's broadcast, the guarded ncclCommInitRank, the rehearsal against
    the unsharded sampler, the timed run, the sharded boundary's loops with their bitwise check, tear-down.  The line
    says it is a rehearsal; what is proved is bench.py's code around the collective, not RCCL."""
    import json, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    side = os.path.join(tempfile.mkdtemp(), "bench_full.json")
    env = dict(os.environ, MBB_BENCH_WALKERS_PER_GPU="32", MBB_BENCH_FULL=side, MBB_RCCL_LIB=_standin_rccl())
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                          "--oversubscribe", "--exchange", "rccl"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] and d["value"] > 0 and d["ranks_agree"] is True, d
    assert d["valid_for_scaling"] is False and "rehearsal" in d["config"]["note"] and len(lines[0]) < 4096
    assert d["exchange_validation"]["rccl"] == "ok", d["exchange_validation"]
    sb = d["boundary_sharded"]
    assert sb["ok"] is True and sb["rows_64"] > 0 and sb["rows_32"] > 0, sb
    full = json.load(open(side))
    assert full["boundary_sharded"]["rows_64"]["equals_unsharded_bitwise"] is True
    assert full["exchange_validation"]["rccl"]["kernel_form"] == 1


def test_a_named_rccl_library_that_does_not_load_is_an_error(mbb, g_lnl):
    """MBB_RCCL_LIB is not a search path: a name that does not load fails the call that needs the communicator (in a
    child process: the library resolves RCCL once per process)."""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mbb_emcee_amd import _native\n"
            "c = _native.Context()\n"
            "try:\n    c.comm_unique_id()\nexcept _native.NativeError as e:\n    print('REFUSED', e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MBB_RCCL_LIB="/nonexistent/librccl.so"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "REFUSED" in out.stdout and "MBB_RCCL_LIB" in out.stdout, out.stdout + out.stderr
