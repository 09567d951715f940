"""A rank of `bench.py --gpus N` that touches nothing and behaves as told (MBB_BENCH_FAKE_WORKER: a comma list of
part, full, noline, crash1, hang1, hang0, fail0), loaded by bench.py through MBB_BENCH_RANK_HOOK: the supervisor's
collecting, merging, waiting and ending of ranks, and the short line, tested without a GPU (tests/test_host_cpu.py)."""
import json
import os
import sys
import time


def run(bench, args):
    how = os.environ["MBB_BENCH_FAKE_WORKER"]
    base_line, emit_line, emit, ROOT = bench.base_line, bench.emit_line, bench.emit, bench.ROOT
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0 and "noline" not in how:
        line = dict(base_line(args, world), value=1.0e6 * world, ms_per_step=1.0)
        if "full" in how:
            # every leg populated: round 4's whole line (20.7 KB, the one the driver could not parse)
            line = dict(json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_v5.json"))), **line)
        emit_line(line)
    under_supervisor = os.environ.get("MBB_BENCH_WORKER") == "1"       # (a one-GPU run has no parts)
    if rank == 0 and "part" in how and under_supervisor:
        emit({"_part": "exchange_validation", "data": {"rccl": {"ok": True, "us_per_step": 20.0}}})
    if rank == 0 and "full" in how and under_supervisor:
        long_why = "rehearsal: the ranks' copies of the ensemble differ after 60 steps " * 20
        emit({"_part": "exchange_validation", "data": {"ipc": {"ok": False, "why": long_why}, "ipc-launches": {"ok": True, "us_per_step": 30.0, "why": long_why},
                                                       "rccl": {"ok": True, "us_per_step": 20.0, "steps": 500, "kernel_form": 1, "why": long_why}}})
        emit({"_part": "boundary_sharded", "data": {"ok": True, "what": long_why, "rows_%d" % (125 * world): {"median_us": 31.123456789, "p90_us": 40.0, "calls": 200},
                                                    "rows_%d" % (250 * world): {"median_us": 35.123456789, "p90_us": 45.0, "calls": 200}}})
    if rank == 1 and "crash1" in how:
        os._exit(7)
    if rank == 1 and "hang1" in how:
        time.sleep(600)
    if rank == 0 and "hang0" in how:
        sys.stdout.flush()
        time.sleep(600)
    if rank == 0 and "fail0" in how:
        os._exit(3)
    return 0
