#!/usr/bin/env python3
"""Registers, scratch and LDS of the library's kernels as the code object declares them.

    python tools/kernel_meta.py [name-substring ...]      # default: solve_m, assoc_group, ghost, chunk_l1

Compiles icm-slam_amd/csrc/icm_api.hip for gfx950 with --save-temps into scratch/isa/ (the .s is what
tools/count_assoc_isa.py and tools/isa_dep_chain.py read) and prints, per kernel, next_free_vgpr / sgpr,
private_segment_fixed_size (scratch bytes per lane) and group_segment_fixed_size (LDS bytes per workgroup)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "scratch", "isa")


def build_isa(defs=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=14", "--save-temps",
           "-c", os.path.join(ROOT, "icm-slam_amd", "csrc", "icm_api.hip"), "-o", os.path.join(OUT, "api.o")] + ["-D" + d for d in defs]
    subprocess.check_call(cmd, cwd=OUT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(OUT, "icm_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(path):
    s = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        body = m.group(2)

        def g(k):
            r = re.search(r"\.amdhsa_" + k + r" (\S+)", body)
            return r.group(1) if r else "?"
        yield m.group(1), g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")


def demangle(n):
    try:
        return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], text=True).strip().split("(")[0]
    except Exception:
        return n


if __name__ == "__main__":
    want = sys.argv[1:] or ["solve_m", "assoc_group", "ghost", "chunk_l1"]
    path = build_isa()
    for name, v, sg, scr, lds in kernels(path):
        if any(w in name for w in want):
            print("%-72s vgpr %4s sgpr %4s scratch %5s B/lane  lds %6s B" % (demangle(name)[:72], v, sg, scr, lds))
