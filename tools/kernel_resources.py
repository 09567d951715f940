#!/usr/bin/env python3
"""Table of registers, spills, scratch and LDS per kernel from the compiler's own remarks.

    python -m mbb_emcee_amd.build --force -v 2> build.log ; python tools/kernel_resources.py build.log

(-Rpass-analysis=kernel-resource-usage, one block of remarks per kernel.)"""
import re
import subprocess
import sys
from collections import OrderedDict


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def parse(path):
    ks, cur = OrderedDict(), None
    for ln in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = ks.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return ks


if __name__ == "__main__":
    ks = parse(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    dm = demangle(list(ks))
    print("%-64s %5s %5s %6s %6s %7s %7s" % ("kernel", "VGPR", "SGPR", "Sspill", "Vspill", "scratch", "LDS"))
    for k, v in ks.items():
        name = re.sub(r"\(.*", "", dm[k]).replace("void ", "")
        if pat and pat not in name:
            continue
        print("%-64s %5d %5d %6d %6d %7d %7d" % (name[:64], v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("SGPRs Spill", -1),
                                               v.get("VGPRs Spill", -1), v.get("ScratchSize", -1), v.get("LDS Size", -1)))
