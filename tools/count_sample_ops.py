#!/usr/bin/env python3
"""Counts the instructions between the markers of tools/count_sample_ops.hip by class."""
import re, sys, json
from collections import Counter
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_hist import classify
txt = open(sys.argv[1]).read().split("\n")
out, cur, name = {}, None, None
for ln in txt:
    m = re.search(r"; MARK_(BEGIN|END) (\w+)", ln)
    if m:
        if m.group(1) == "BEGIN":
            cur, name = Counter(), m.group(2)
        else:
            out[name] = cur; cur = None
        continue
    s = ln.strip()
    if cur is None or not s or s.startswith((";", ".", "//")):
        continue
    op = s.split()[0]
    cur[classify(op)] += 1
for k, c in out.items():
    valu = sum(v for kk, v in c.items() if kk.startswith(("f64", "v_", "dpp")))
    print("%-18s VALU %3d  (fp64 fma/mul/add %3d, other fp64 %2d, 32-bit %2d)  LDS reads %d" %
          (k, valu, c.get("f64_fma_mul_add", 0), c.get("f64_other", 0),
           valu - c.get("f64_fma_mul_add", 0) - c.get("f64_other", 0), c.get("lds", 0)))
if len(sys.argv) > 2:
    json.dump({k: dict(v) for k, v in out.items()}, open(sys.argv[2], "w"), indent=1)
