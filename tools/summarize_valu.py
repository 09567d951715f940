#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_INSTS_VALU ... counter_collection CSV -> profiles/rNN/pmc_valu_*.json:
medians per launch for every k_lnlike / k_flowm instantiation in the run, largest grid only
(a run of bench.py or tools/bench_cfg5.py also launches small set-up batches).

    python tools/summarize_valu.py <rocprof output dir> <out.json> "<command that was profiled>" [note] [half_steps]

half_steps (a number, or the log of the profiled bench.py run): for the one-launch sampler kernels (k_lnlike<.., 5, ..>, k_flowm<..>), whose launches cover different
numbers of half-steps, the number of half-steps all its launches in the run add up to; the
summary then also carries the counters summed over the launches and per half-step.
"""
import csv, glob, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _half_steps import half_steps as _hs


def main(run_dir, out, command="", note="", half_steps="0"):
    path = glob.glob(run_dir + "/*/*_counter_collection.csv")[0]
    per = {}
    for r in csv.DictReader(open(path)):
        if not any(n in r["Kernel_Name"] for n in ("k_lnlike<", "k_flowm<", "k_flowa<")):
            continue
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        per.setdefault(r["Kernel_Name"], {}).setdefault(grid, {}).setdefault(r["Counter_Name"], []).append(
            float(r["Counter_Value"]))
    res = {"command": command, "note": note, "kernels": {}}
    for k, grids in per.items():
        # the launch shape that was dispatched most often is the timed one
        g = max(grids, key=lambda gg: len(next(iter(grids[gg].values()))))
        c = {cn: float(np.median(v)) for cn, v in grids[g].items()}
        fma, add, mul = (c.get("SQ_INSTS_VALU_" + x, 0.0) for x in ("FMA_F64", "ADD_F64", "MUL_F64"))
        f64 = fma + add + mul + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        extra = {}
        if any(n in k for n in ("k_flowm<", "k_flowa<")) and _hs(half_steps) > 0:
            tot = {cn: float(np.sum(v)) for cn, v in grids[g].items()}
            extra = {"half_steps_in_all_launches": _hs(half_steps), "counters_all_launches": tot,
                     "counters_per_half_step": {cn: v / _hs(half_steps) for cn, v in tot.items()}}
        res["kernels"][k] = {
            "grid_threads": g, "dispatches": len(next(iter(grids[g].values()))),
            "counters_per_launch": c, "fp64_wave_instructions": f64,
            "fp64_share_of_valu": f64 / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None,
            "fp64_flops_per_launch": 64.0 * (2 * fma + add + mul), **extra}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
