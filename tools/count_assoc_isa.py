#!/usr/bin/env python3
"""Instruction mix of k_assoc_group (the sweep's dominant kernel) from its gfx950 ISA, per basic block.

    python tools/kernel_meta.py >/dev/null      # writes scratch/isa/*.s (hipcc --save-temps)
    python tools/count_assoc_isa.py [kernel-name-prefix]

Per basic block: vector instructions split into FP64 arithmetic (4 issue cycles per wave64 instruction: 16 FP64 lanes
per SIMD per cycle), DPP moves / 32-bit and integer vector ops (2 cycles with another wave ready, 4 for a lone wave:
MI355X_MICROARCH.md, per-instruction cycle constants), scalar, LDS and memory instructions, and where the block
branches.  The batch loop (one iteration per 64 beams) is the block chain between the loop label and its back edge;
DESIGN.md section 5 prices it against GRBM_GUI_ACTIVE."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "scratch", "isa", "icm_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_lines(prefix):
    lines = open(PATH).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start + 1:end + 1]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "_f64" in op or op in ("v_fmac_f64", "v_fma_f64"):
            return "valu_f64"
        if op.endswith("_dpp") or "dpp" in ins or "row_" in ins or "quad_perm" in ins or "wave_sh" in ins:
            return "valu_dpp"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        return "valu_32"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "_ZN3icm13k_assoc_groupILb0ELb0ELi128ELi1ELi4EEE"
    cur, blocks = "entry", collections.OrderedDict({"entry": []})
    for l in kernel_lines(prefix):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.split(";")[0].strip()
        if t and not t.startswith("."):
            blocks[cur].append(t)
    kinds = ("valu_f64", "valu_32", "valu_dpp", "valu_lane", "salu", "lds", "vmem", "wait", "branch")
    print("%-10s %5s " % ("block", "n") + " ".join("%8s" % k for k in kinds) + "  -> branches")
    tot = collections.Counter()
    for name, ins in blocks.items():
        c = collections.Counter(classify(i) for i in ins)
        tot.update(c)
        br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        print("%-10s %5d " % (name, len(ins)) + " ".join("%8d" % c[k] for k in kinds) + "  -> " + ",".join(br))
    print("%-10s %5d " % ("total", sum(tot.values())) + " ".join("%8d" % tot[k] for k in kinds))


if __name__ == "__main__":
    main()
