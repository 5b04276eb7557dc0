"""Back-to-back dependence statistics of a kernel's VALU stream (gfx950 .s from --save-temps).
A lone wavefront issues a VALU instruction that depends on the one right before it every ~8.4 cycles and an
independent one every ~5.5 cycles (tools/ubench_issue.hip), so the share of dependent neighbours in a
latency-bound kernel's hot loop is its headroom from instruction scheduling alone.
usage: isa_dep_chain.py file.s kernel-substring [first_line last_line]"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b', tok):
        if m.group(1):
            out.update((m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    if 'vcc' in tok:
        out.add(('vcc', 0))
    return out


def analyse(lines):
    prev_defs = set()
    n = dep = 0
    for l in lines:
        l = l.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.'):
            continue
        op = l.split()[0]
        if not op.startswith('v_'):
            if op.startswith(('s_cbranch', 's_branch')):
                prev_defs = set()
            continue
        args = l[len(op):].split(',')
        ndst = 2 if (op.startswith(('v_div_scale', 'v_add_co', 'v_sub_co', 'v_addc', 'v_subb', 'v_mad_u64', 'v_mad_i64'))) else 1
        if op.startswith('v_cmp'):
            dst = regs(args[0]) if not op.endswith('_e32') else {('vcc', 0)}
            src = set().union(*[regs(a) for a in args[1:]]) if len(args) > 1 else set()
            if op.endswith('_e32') or len(args) == 2:
                dst, src = {('vcc', 0)}, set().union(*[regs(a) for a in args])
        else:
            dst = set().union(*[regs(a) for a in args[:ndst]])
            src = set().union(*[regs(a) for a in args[ndst:]]) if len(args) > ndst else set()
            if op.startswith(('v_fmac', 'v_mac')):
                src |= dst
            if op.startswith('v_cndmask') and len(args) == 3:
                src.add(('vcc', 0))
        n += 1
        if src & prev_defs:
            dep += 1
        prev_defs = dst
    return n, dep


if __name__ == '__main__':
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(\S*%s\S*):' % re.escape(key), s, re.M)
    i = m.start()
    j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    a, b = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
    n, dep = analyse(body[a:b])
    print('%s lines %d-%d: %d VALU, %d (%.0f%%) depend on the VALU right before -> est. %.0f cycles lone-wave (8.4 dep / 5.5 indep)' % (
        m.group(1)[:60], a, b, n, dep, 100.0 * dep / max(n, 1), dep * 8.4 + (n - dep) * 5.5))
