import csv, glob, sys
import numpy as np
path = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
rows = {}
for r in csv.DictReader(open(path)):
    if "lnlike" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
        rows.setdefault((int(r["Grid_Size"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k in sorted(rows):
    v = np.array(rows[k])
    print("grid threads %7d  %s: median %.2f KB, min %.2f, max %.2f (%d launches)" % (k[0], k[1], np.median(v), v.min(), v.max(), len(v)))
