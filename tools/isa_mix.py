#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing:  tools/isa_mix.py file.s <symbol-substring> [--loop]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
body = m.group(2)
lines = [l.strip() for l in body.split('\n')]
lines = [l for l in lines if l and not l.startswith(('.', ';', '//'))]
if '--loop' in sys.argv:
    # keep the largest backward-branch loop body
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    best = (0, 0)
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\S+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] > best[1] - best[0]:
            best = (labels[mm.group(1)], i)
    lines = lines[best[0]:best[1] + 1]
c = collections.Counter(l.split()[0] for l in lines if not l.endswith(':'))
g = collections.Counter()
for k, v in c.items():
    if k.startswith('v_pk'): g['v_pk'] += v
    elif k.startswith('v_'): g['valu'] += v
    elif k.startswith('ds_'): g['ds'] += v
    elif k.startswith(('global_', 'buffer_', 'flat_')): g['vmem'] += v
    elif k.startswith('s_waitcnt'): g['waitcnt'] += v
    elif k.startswith('s_'): g['salu'] += v
    else: g[k] += v
print('total', sum(c.values()), dict(g))
print(c.most_common(45))
