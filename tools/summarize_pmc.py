#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two
separate passes, MI355X_MICROARCH.md 'rocprofv3 PMC slots') into a small JSON
summary under profiles/.  Values are KB per dispatch as rocprofv3 reports them."""
import csv, glob, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _half_steps import half_steps as _hs

def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"n": len(v), "median_KB": float(np.median(v)), "mean_KB": float(np.mean(v)), "sum_KB": float(np.sum(v))}
            for k, v in out.items()}

def main(fetch_dir, write_dir, out, half_steps="0"):
    f = per_kernel(glob.glob(fetch_dir + "/*/*_counter_collection.csv")[0], "FETCH_SIZE")
    w = per_kernel(glob.glob(write_dir + "/*/*_counter_collection.csv")[0], "WRITE_SIZE")
    res = {"units": "KB per dispatch as reported by rocprofv3 (FETCH_SIZE = TCC_EA0_RDREQ x 64 B)",
           "note": "gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by 2x; the "
                   "staged kernel reads its tables that way (calibration: tools/probe_traffic_calib.py), "
                   "so add 8 XCD copies x table bytes / 2 to traffic_bytes_per_launch. "
                   "Infinity-Cache hits are counted.",
           "kernels": {}}
    for k in f:
        if "lnlike" in k or "k_flow" in k:
            res["kernels"][k] = {"FETCH_SIZE": f[k], "WRITE_SIZE": w.get(k),
                                 "traffic_bytes_per_launch": 1024.0 * (f[k]["median_KB"] + (w[k]["median_KB"] if k in w else 0.0))}
            # half_steps: a number, or "<log of the FETCH pass>,<log of the WRITE pass>" (the passes are separate runs)
            hs = [_hs(x) for x in (str(half_steps).split(",") * 2)[:2]]
            if any(n in k for n in ("k_flowm", "k_flowa")) and min(hs) > 0:
                # the one-launch sampler kernel: its launches cover different numbers of half-steps
                res["kernels"][k]["half_steps_in_all_launches"] = {"fetch_pass": hs[0], "write_pass": hs[1]}
                res["kernels"][k]["traffic_bytes_per_half_step"] = \
                    1024.0 * (f[k]["sum_KB"] / hs[0] + (w[k]["sum_KB"] / hs[1] if k in w else 0.0))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:5])
