#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_INSTS_VALU ... counter_collection CSV of `tools/bench_cfg5.py --quick`
-> profiles/rNN/pmc_valu_cfg5_*.json (medians over the 250 000-walker launches)."""
import csv, glob, json, sys
import numpy as np

def main(run_dir, out, note=""):
    path = glob.glob(run_dir + "/*/*_counter_collection.csv")[0]
    per = {}
    for r in csv.DictReader(open(path)):
        if "k_lnlike<false, false, false," in r["Kernel_Name"]:       # thick, alpha, plain
            per.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    # the variant with the largest instruction count is the cfg5 launch
    kern = max({k for k, _ in per}, key=lambda k: np.median(per.get((k, "SQ_INSTS_VALU"), [0])))
    c = {cn: float(np.median(v)) for (k, cn), v in per.items() if k == kern}
    f64 = c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)
    flops = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0))
    res = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 "
                      "SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace "
                      "-- python3 tools/bench_cfg5.py --quick",
           "kernel": kern, "note": note, "counters_per_launch": c,
           "fp64_wave_instructions": f64, "fp64_share_of_valu": f64 / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None,
           "fp64_flops_per_launch": flops,
           "valu_wave_instructions_per_walker": c.get("SQ_INSTS_VALU", 0) / 250000.0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:4])
