#!/usr/bin/env python3
"""Where a kernel's spilled scalar registers are written and read back, by source line.

    hipcc ... -gline-tables-only -S --cuda-device-only -o k.s file.hip
    python tools/spill_sites.py k.s <mangled kernel name>

SGPR spills live in lanes of a VGPR: v_writelane_b32 where the value is put away, v_readlane_b32
where it comes back -- the read-backs inside a loop are the ones that cost (they sit in front of
every use, each iteration)."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^%s:(.*?)s_endpgm" % re.escape(name), txt, flags=re.S | re.M)
files = dict((int(a), b.split("/")[-1]) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt))
files.update(dict((int(a), b.split("/")[-1]) for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', txt, flags=re.M)))
cur = ("?", 0)
wr, rd = Counter(), Counter()
for ln in m.group(1).split("\n"):
    s = ln.strip()
    mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if mm:
        cur = (files.get(int(mm.group(1)), mm.group(1)), int(mm.group(2)))
        continue
    if s.startswith("v_writelane_b32"):
        wr[cur] += 1
    elif s.startswith("v_readlane_b32") and not "dpp" in s:
        rd[cur] += 1
print("written (v_writelane): %d" % sum(wr.values()))
for k, v in sorted(wr.items()):
    print("   %s:%d  x%d" % (k[0], k[1], v))
print("read back (v_readlane): %d" % sum(rd.values()))
for k, v in sorted(rd.items()):
    print("   %s:%d  x%d" % (k[0], k[1], v))
