#!/usr/bin/env python3
"""Dev tool: instruction mix of one kernel's hottest loop from hipcc -S output (issue-cost weighted per the guide's
constants: transcendental 8 cycles, MFMA 8, other VALU 4).  usage: isa_mix.py file.s kernel_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
start = re.search(r'^(\S*' + re.escape(key) + r'\S*):\s*(;.*)?$', s, flags=re.M)
end = s.index('.Lfunc_end', start.end())
body = s[start.end():end].split('\n')
print(start.group(1), len(body), "lines")
# basic blocks
blocks, cur, name = [], [], "entry"
for ln in body:
    t = ln.strip()
    if re.match(r'^\.LBB\S+:', t):
        blocks.append((name, cur)); name, cur = t[:-1], []
    elif t and not t.startswith(('.', ';', '//')):
        cur.append(t.split()[0])
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'
big = sorted(blocks, key=lambda b: -sum(1 for o in b[1] if o.startswith('v_mfma')))[:4]
for name, ops in big:
    c = collections.Counter(cls(o) for o in ops)
    issue = 8 * c['mfma'] + 8 * c['trans'] + 4 * c['valu']
    print(f"{name}: {len(ops)} instrs {dict(c)}  vector issue cycles ~{issue} (mfma pipe {32 * c['mfma']})")
    v = collections.Counter(o for o in ops if cls(o) == 'valu')
    print("   ", v.most_common(14))
