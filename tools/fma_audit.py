#!/usr/bin/env python3
"""Per kernel: how many fp64 fma / mul / add instructions the device code holds, for two builds
of the same translation unit (default -ffp-contract against -ffp-contract=off).  A kernel whose
counts differ holds a product-and-sum the compiler contracted on its own: that site wants an
explicit fma() (or wants to stay two roundings) before the library is built with contraction off.

    python tools/fma_audit.py a.s b.s
"""
import re
import sys
from collections import Counter, OrderedDict


def per_kernel(path):
    out, cur = OrderedDict(), None
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = out.setdefault(m.group(1), Counter())
            continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith("s_endpgm"):
            cur = None
            continue
        op = s.split()[0] if s else ""
        if op.startswith(("v_fma_f64", "v_fmac_f64")):
            cur["fma"] += 1
        elif op.startswith("v_mul_f64"):
            cur["mul"] += 1
        elif op.startswith("v_add_f64"):
            cur["add"] += 1
    return out


if __name__ == "__main__":
    a, b = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    same = 0
    for k in a:
        if k not in b:
            continue
        if a[k] == b[k]:
            same += 1
            continue
        print("%-70s fma %5d -> %5d   mul %5d -> %5d   add %5d -> %5d" %
              (k[:70], a[k]["fma"], b[k]["fma"], a[k]["mul"], b[k]["mul"], a[k]["add"], b[k]["add"]))
    print("%d kernels with identical fp64 fma/mul/add counts, %d in all" % (same, len(a)))
