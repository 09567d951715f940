"""Side channel for the multi-process tests: ranks started by the test itself (subprocess.Popen, RANK / WORLD_SIZE /
MBB_TEST_RDZV_DIR in the environment) meet through files in a directory the parent made.  No port is ever chosen, so
nothing can find one taken; no torch, no launcher.  Every message is a file written under a temporary name and
renamed into place (rename is atomic: a reader sees the whole message or none).

It speaks the two protocols mbb_emcee_amd.parallel accepts -- ``allgather_bytes`` + ``barrier`` (ipc_exchange_setup's
side channel) and ``rank`` / ``world`` / ``allgather_host`` (ShardedLikelihood's host communicator) -- and starts the
ranks (`launch`)."""
import os
import pickle
import subprocess
import sys
import time

import numpy as np


class FileComm(object):
    def __init__(self, directory=None, rank=None, world=None, timeout=120.0):
        self.dir = directory or os.environ["MBB_TEST_RDZV_DIR"]
        self.rank = int(os.environ["RANK"] if rank is None else rank)
        self.world = int(os.environ["WORLD_SIZE"] if world is None else world)
        self.timeout = float(timeout)
        self._seq = 0

    def _path(self, seq, rank):
        return os.path.join(self.dir, "m%06d.%d" % (seq, rank))

    def allgather_bytes(self, b):
        """Every rank's bytes, in rank order, on every rank."""
        seq, self._seq = self._seq, self._seq + 1
        tmp = self._path(seq, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(bytes(b))
        os.rename(tmp, self._path(seq, self.rank))
        out, t_end = [], time.time() + self.timeout
        for r in range(self.world):
            p = self._path(seq, r)
            while not os.path.exists(p):
                if time.time() > t_end:
                    raise TimeoutError("rank %d: rank %d never reached step %d of the side channel" % (self.rank, r, seq))
                if os.path.exists(os.path.join(self.dir, "abort")):
                    raise RuntimeError("rank %d: another rank gave up" % self.rank)
                time.sleep(0.0005)
            with open(p, "rb") as f:
                out.append(f.read())
        return out

    def barrier(self):
        self.allgather_bytes(b"")

    def allgather_object(self, obj):
        return [pickle.loads(b) for b in self.allgather_bytes(pickle.dumps(obj))]

    def allgather_host(self, x):
        """ShardedLikelihood's host communicator: the concatenation over ranks of equal-length float64 vectors."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        return np.concatenate([np.frombuffer(b, dtype=np.float64) for b in self.allgather_bytes(x.tobytes())])

    def abort(self):
        open(os.path.join(self.dir, "abort"), "w").close()


def launch(script, world, extra_env=None, args=(), timeout=300.0):
    """Start `world` ranks of `script` as children of this process and wait for them.  Returns (ok, transcript): ok when
    every rank left with status 0.  A rank that fails marks the directory so that the others stop waiting; on the
    deadline exactly the children started here are killed."""
    import tempfile
    d = tempfile.mkdtemp(prefix="mbb_rdzv_")
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "MASTER_ADDR", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", MBB_TEST_RDZV_DIR=d, WORLD_SIZE=str(world))
    env.update(extra_env or {})
    logs = [open(os.path.join(d, "rank%d.log" % r), "wb") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, script] + list(args), env=dict(env, RANK=str(r)), stdout=logs[r],
                              stderr=subprocess.STDOUT) for r in range(world)]
    t_end = time.time() + timeout
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) and not os.path.exists(os.path.join(d, "abort")):
            open(os.path.join(d, "abort"), "w").close()
        if time.time() > t_end:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                  # exactly the processes started above
            break
        time.sleep(0.02)
    for p in procs:
        p.wait()
    for f in logs:
        f.close()
    text = ""
    for r in range(world):
        with open(os.path.join(d, "rank%d.log" % r), "rb") as f:
            text += "---- rank %d (status %s)\n%s\n" % (r, procs[r].returncode, f.read().decode(errors="replace"))
    return all(p.returncode == 0 for p in procs), text
