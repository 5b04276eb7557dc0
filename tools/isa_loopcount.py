#!/usr/bin/env python3
"""Instruction counts of the loops of one kernel in a device assembly file.

    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S -o /tmp/api.s icm-slam_amd/csrc/icm_api.hip
    python tools/isa_loopcount.py /tmp/api.s _ZN3icm15k_solve_m_fusedILb0ELb1EEE

Every back edge of the kernel's control-flow graph with more than 100 instructions between its target and itself: vector
instructions (FP64 arithmetic, moves, selects), scalar ones, padding s_nop.  DESIGN.md quotes the Nelder-Mead iteration of
the fold-only solve from it (286 -> 211 instructions in round 3)."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^' + pat + r'.*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i] and i > start + 50)
# a kernel may have several s_endpgm; take up to .Lfunc_end
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
lab, ins = {}, []
meta = {}
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab[m.group(1)] = len(ins)
    m = re.match(r'^\s+([vs]_[a-z0-9_]+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)\s*(.*)', l)
    if m: ins.append((m.group(1), m.group(2)))
for l in lines[end:end + 80]:
    m = re.match(r'\s*;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy): (\d+)', l)
    if m: meta[m.group(1)] = int(m.group(2))
print(len(ins), 'instrs', meta)
loops = []
for k, (op, args) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        t = args.split()[0]
        if t in lab and lab[t] <= k:
            loops.append((lab[t], k, t))
for a, b, t in loops:
    body = ins[a:b + 1]
    v = sum(1 for o, _ in body if o.startswith('v_'))
    mv = sum(1 for o, _ in body if o.startswith('v_mov'))
    cnd = sum(1 for o, _ in body if o.startswith('v_cndmask'))
    f64 = sum(1 for o, _ in body if '_f64' in o)
    s = sum(1 for o, _ in body if o.startswith('s_'))
    nop = sum(1 for o, _ in body if o == 's_nop')
    if len(body) > 100:
        print(f'loop {t} [{a}..{b}] len {len(body)} valu {v} (f64 {f64} mov {mv} cndmask {cnd}) salu {s} (nop {nop})')
