#!/usr/bin/env python3
"""instruction mix of the MFMA-heaviest loop of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only)"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for (i, name) in starts:
    j = i
    while j < len(lines) and 's_endpgm' not in lines[j]: j += 1
    body = lines[i:j]
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = k
    best = None
    for k, l in enumerate(body):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            n = sum(1 for x in body[labels[m.group(1)]:k] if 'v_mfma' in x)
            if best is None or n > best[0]: best = (n, labels[m.group(1)], k)
    if not best: continue
    n, a, b = best
    cnt = {}
    for x in body[a:b]:
        x = x.strip()
        if not x or x.startswith(('.', ';')): continue
        op = x.split()[0]
        key = ('mfma' if 'mfma' in op else 'ds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_'))
               else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other')
        cnt[key] = cnt.get(key, 0) + 1
    print(name.split(':')[0][:70], 'loop lines', b - a, cnt)
