#!/usr/bin/env python3
"""Histogram of the gfx950 ISA of one kernel by issue class, per basic block.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o mbb.s mbb_emcee_amd/csrc/mbb_hip.hip
    python tools/isa_hist.py mbb.s _Z8k_lnlikeILb0ELb0ELb0ELb0EEv8LikeArgs [--blocks]

Classes follow what tools/lat_ops.hip measures: fp64 FMA/MUL/ADD (4-cycle issue on a
16-lane-per-clock fp64 pipe), other fp64 (rcp, ldexp, cmp, min/max, cvt, rndne), 32-bit VALU
(int/logic/cndmask/mov), DPP moves, LDS, vector memory, scalar ALU/memory, branches.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
        return "f64_fma_mul_add"
    if op.startswith("v_") and "f64" in op:
        return "f64_other"
    if op.startswith("v_") and ("dpp" in op):
        return "dpp"
    if op.startswith(("v_cndmask",)):
        return "v_cndmask"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_lane"
    if op.startswith("v_cmp"):
        return "v_cmp32"
    if op.startswith("v_"):
        return "v_int32_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
            continue
        if on:
            out.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    return out


def blocks(lines):
    bl = OrderedDict()
    cur = "entry"
    bl[cur] = []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            bl[cur] = []
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        # v_xxx_dpp forms are spelled with a trailing modifier in the operand list
        if "row_" in s or "quad_perm" in s or "row_newbcast" in s:
            op = op + "_dpp"
        bl[cur].append((op, s))
    return bl


def main():
    path, name = sys.argv[1], sys.argv[2]
    show = "--blocks" in sys.argv
    bl = blocks(kernel_lines(path, name))
    total = Counter()
    order = list(bl.keys())
    # back edges: a block that branches to a label at or above itself
    idx = {k: i for i, k in enumerate(order)}
    loops = []
    for k, ins in bl.items():
        for op, s in ins:
            if op.startswith(("s_cbranch", "s_branch")):
                t = s.split()[-1]
                if t in idx and idx[t] <= idx[k]:
                    loops.append((t, k))
    for k, ins in bl.items():
        c = Counter(classify(op) for op, _ in ins)
        total.update(c)
        if show and ins:
            print("%-10s n=%4d  " % (k, len(ins)) + " ".join("%s=%d" % kv for kv in sorted(c.items())))
    print("TOTAL", sum(total.values()), dict(sorted(total.items())))
    print("LOOPS (head <- tail):", loops)
    minlen = 0
    for a in sys.argv:
        if a.startswith("--min="):
            minlen = int(a[6:])
    for head, tail in loops:
        if sum(len(bl[k]) for k in order[idx[head]:idx[tail] + 1]) < minlen:
            continue
        c = Counter()
        ops = Counter()
        for k in order[idx[head]:idx[tail] + 1]:
            c.update(classify(op) for op, _ in bl[k])
            ops.update(op for op, _ in bl[k])
        valu = sum(v for kk, v in c.items() if kk.startswith(("f64", "v_", "dpp")))
        print("loop %s..%s: %d instr, %d VALU: %s" % (head, tail, sum(c.values()), valu, dict(sorted(c.items()))))
        if "--ops" in sys.argv:
            print("   ", ops.most_common(60))


if __name__ == "__main__":
    main()
