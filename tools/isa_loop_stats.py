#!/usr/bin/env python3
"""Instruction mix of the hottest loop of one kernel in a hipcc -S listing.
    python tools/isa_loop_stats.py file.s <substring of mangled kernel name>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
print(m.group(1))
body = m.group(2).split('\n')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
loops = []
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))
lo, hi = max(loops, key=lambda x: x[1] - x[0])
loop = [l for l in body[lo:hi + 1] if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
c = Counter(l.split()[0] for l in loop)
valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
print(f"kernel {len(body)} lines; hottest loop {len(loop)} instrs: VALU {valu}, MFMA {sum(v for k, v in c.items() if 'mfma' in k)}, "
      f"DS {sum(v for k, v in c.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_')))}, "
      f"s_waitcnt {c.get('s_waitcnt', 0)}, s_barrier {c.get('s_barrier', 0)}")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f"{v:5d} {k}")
