#!/usr/bin/env python3
"""Count VALU / scratch instructions per backward-branch loop body in a hipcc -save-temps .s file."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):\s*;\s*@\1\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    if pat not in name:
        continue
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    tot = sum(1 for x in body if re.match(r'^\s+v_', x))
    print(f'{name}: total VALU {tot}')
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = body[labels[mm.group(1)]:i]
            v = [x.split()[0] for x in seg if re.match(r'^\s+v_', x)]
            sc = [x for x in seg if 'scratch_' in x]
            print(f'  loop {mm.group(1)}: valu {len(v)} scratch {len(sc)}', Counter(v).most_common(10))
