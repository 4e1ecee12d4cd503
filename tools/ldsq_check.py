#!/usr/bin/env python3
"""Replays the LDS return queue over a gfx950 assembly listing (one kernel, straight-line order) and flags instructions that
read or overwrite a VGPR whose ds_read is still outstanding according to the s_waitcnt lgkmcnt(...) instructions seen so far.
LDS operations retire in order; inline-asm reads are invisible to hipcc's own wait insertion, this is the cross-check.
usage: ldsq_check.py file.s [kernel-name-substring]"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2] if len(sys.argv) > 2 else None
def regs(tok):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
on = sub is None
q = []   # outstanding LDS ops: (line, dest regs)
flags = 0
for i, l in enumerate(L):
    t = l.strip()
    if re.match(r'^_Z\w+:', l):
        on = (sub is None) or (sub in l); q = []
        continue
    if not on or not t or t[0] in '.;': continue
    if 's_endpgm' in t: on = sub is None; continue
    op = t.split()[0]
    if op == 's_waitcnt':
        m = re.search(r'lgkmcnt\((\d+)\)', t)
        if m:
            n = int(m.group(1))
            while len(q) > n: q.pop(0)
        continue
    if op == 's_barrier': continue
    ops = t.split(None, 1)[1] if ' ' in t else ''
    toks = [x.strip().split()[0] for x in ops.split(',') if x.strip()]
    if op.startswith('ds_'):
        # sources: all operands for writes, operands after the first for reads
        isread = op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle'))
        src = set()
        for x in (toks[1:] if isread else toks): src |= regs(x)
        dst = regs(toks[0]) if isread else set()
        for (ln, d) in q:
            if d & src: flags += 1; print(f"line {i+1}: {t}   reads v{sorted(d & src)} of outstanding LDS op at line {ln+1}")
            if d & dst: flags += 1; print(f"line {i+1}: {t}   overwrites dest of outstanding LDS op at line {ln+1}")
        q.append((i, dst))
        continue
    if op.startswith(('v_', 'buffer_', 'global_', 'scratch_')):
        src = set(); dst = set()
        if op.startswith(('buffer_store', 'global_store', 'scratch_store', 'global_load_lds')):
            for x in toks: src |= regs(x)
        else:
            dst = regs(toks[0]) if toks else set()
            for x in toks[1:]: src |= regs(x)
            if 'mfma' in op or op.startswith('v_fmac') or op.startswith('v_pk_fma') is False and False: pass
        for (ln, d) in q:
            if d & src: flags += 1; print(f"line {i+1}: {t[:90]}   reads v{sorted(d & src)} of outstanding LDS op at line {ln+1}: {L[ln].strip()}")
            elif d & dst: flags += 1; print(f"line {i+1}: {t[:90]}   overwrites v{sorted(d & dst)}, dest of outstanding LDS op at line {ln+1}: {L[ln].strip()}")
print("flags", flags)
