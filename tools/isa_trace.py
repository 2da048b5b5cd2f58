#!/usr/bin/env python3
"""Dev tool: one letter per instruction for the MFMA-bearing blocks of a kernel in hipcc -S output.
M mfma, t transcendental, v other VALU, a v_accvgpr_*, L LDS, G LDS-DMA, g other global, S scratch (spill), w s_waitcnt,
B barrier, n s_nop, . other scalar, > branch.   usage: isa_trace.py file.s kernel_substring [min_mfma]"""
import re
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
start = re.search(r'^(\S*' + re.escape(key) + r'\S*):\s*(;.*)?$', s, flags=re.M)
end = s.index('.Lfunc_end', start.end())
body = s[start.end():end].split('\n')
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith(('v_exp', 'v_log', 'v_rcp')): return 't'
    if op.startswith('v_accvgpr'): return 'a'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_load_lds', 'buffer_')): return 'G'
    if op.startswith('scratch_'): return 'S'
    if op.startswith(('global_', 'flat_')): return 'g'
    if op.startswith('s_waitcnt'): return 'w'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith(('s_cbranch', 's_branch')): return '>'
    if op.startswith('s_nop'): return 'n'
    if op.startswith('s_'): return '.'
    return '?'
out, name, cur = [], 'entry', ''
for ln in body:
    t = ln.strip()
    if re.match(r'^\.LBB\S+:', t):
        out.append((name, cur)); name, cur = t.split(':')[0], ''
    elif t and not t.startswith(('.', ';', '//')):
        cur += cls(t.split()[0])
out.append((name, cur))
tot = ''.join(c for _, c in out)
print(start.group(1)[:80], "total:", {k: tot.count(k) for k in 'MtvaLGSwBn'})
for n, c in out:
    if c.count('M') >= minm:
        print(n, len(c), {k: c.count(k) for k in 'MtvaLS'})
        print(c)
