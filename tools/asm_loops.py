#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing: tools/asm_loops.py file.s <kernel-name-substring> [min_instrs]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):[^\n]*\n", s, re.M)
i = m.end()
body = s[i:s.index(".Lfunc_end", i)].split('\n')
labels = {}
for n, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = n
seen = set()
for n, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        a = labels[mm.group(1)]
        if a in seen:
            continue
        ins = [x.split()[0] for x in body[a:n] if x.startswith('\t') and not x.startswith('\t.') and not x.startswith('\t;')]
        if len(ins) < minn:
            continue
        seen.add(a)
        c = Counter()
        for x in ins:
            k = 'valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else 'lds' if x.startswith('ds_') else 'vmem' if x.startswith(('buffer_', 'global_', 'flat_')) else 'other'
            c[k] += 1
        top = Counter(ins).most_common(14)
        print(mm.group(1), 'lines', a, n, dict(c))
        print('   ', top)
