#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two
separate passes, MI355X_MICROARCH.md 'rocprofv3 PMC slots') into a small JSON
summary under profiles/.  Values are KB per dispatch as rocprofv3 reports them."""
import csv, glob, json, sys
import numpy as np

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
            if any(n in k for n in ("k_flowm", "k_flowa", "k_flowr")) and int(half_steps) > 0:
                # the one-launch sampler kernel: its launches cover different numbers of half-steps
                res["kernels"][k]["half_steps_in_all_launches"] = int(half_steps)
                res["kernels"][k]["traffic_bytes_per_half_step"] = \
                    1024.0 * (f[k]["sum_KB"] + (w[k]["sum_KB"] if k in w else 0.0)) / int(half_steps)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:5])
