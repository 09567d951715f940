#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extras [--steps K --warmup W]` -> the dominant
kernel's duration per half-step, launch by launch, next to the live HIP-event figure of the same run.

    python tools/summarize_stats.py <rocprof output dir> <bench line json> <out.json>

The one-launch sampler kernel covers a different number of half-steps in each of its launches (the 60-step
rehearsal, the warm-up, the timed region), so "average duration per launch" means nothing for it: the
summary gives every launch's duration divided by the half-steps it covered, and the timed one beside the
line's `half_step_us` (HIP events over the timed region)."""
import csv
import glob
import json
import sys


def main(run_dir, bench_json, out):
    line = json.load(open(bench_json))
    steps, warm = line["steps"], line["warmup"]
    trace = glob.glob(run_dir + "/*/*_kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(trace)) if any(n in r["Kernel_Name"] for n in ("k_flowm<", "k_flowa<"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # bench.py says which sampler runs it made (rehearsal, the scratch ensemble that preconditions the GPU, warm-up, the
    # timed region, eight more of it); runs of fewer than two steps are launch trains, not launches of this kernel
    runs = line.get("sampler_runs") or {"steps_of_every_run": [60, warm, steps], "timed_run": 2}
    covered = [st for st in runs["steps_of_every_run"] if st >= 2]
    timed = sum(1 for st in runs["steps_of_every_run"][:runs["timed_run"]] if st >= 2)
    launches = []
    for r, st in zip(rows, covered):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        launches.append({"kernel": r["Kernel_Name"], "steps": st, "duration_us": dur, "us_per_half_step": dur / (2 * st),
                         "workgroups": int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                         # (the trace's VGPR_Count column as it is: on this target it is HALF the compiler's figure -- k_flowm<thick,alpha>:
                         # 52 here, 103 in the resource remarks, i.e. 104 allocated: profiles/rNN/kernel_resources.txt is the
                         # number to quote -- round 5's file called this field "vgprs")
                         "vgpr_count_column_of_the_trace": int(r["VGPR_Count"]),
                         "sgprs": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                         "scratch_bytes": int(r["Scratch_Size"])})
    res = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-extras --steps %d --warmup %d" % (steps, warm),
           "launches_of_the_dominant_kernel": launches, "launches_seen": len(rows),
           "timed_launch": timed,
           "timed_launch_us_per_half_step_rocprof": launches[timed]["us_per_half_step"] if len(launches) == len(covered) == len(rows) else None,
           "half_step_us_hip_events_same_run": line.get("half_step_us"),
           "value_same_run": line.get("value")}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
