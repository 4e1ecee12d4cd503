#!/usr/bin/env python
"""Per-workgroup lifetimes of dhconv_strip.hip (s_memtime stamps of wave 0).  Needs a library built with -DACE_DH_TRACE:
tools/mkvar.sh dhtrace -DACE_DH_TRACE; ACE_SFNO_LIB=exp/libexp_dhtrace.so python tools/trace_dh.py"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ace_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
bench.ACE2 = dict(bench.ACE2, num_layers=1)      # ONE block: one dhconv launch per forward
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision("f16x3")
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net(x)
torch.cuda.synchronize()
L = _lib.lib()
N = 8192
spans = (ctypes.c_ulonglong * (N * 6))()
fn = L.ace_debug_dh_spans
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(spans) == 0
per = collections.defaultdict(list)
cus = collections.defaultdict(list)   # (XCC, cu | sh | se bits of HW_ID) -> its workgroups
for b in range(N):
    s0, s1, s2, l, xcc, hwid = (spans[6 * b + k] for k in range(6))
    if s2 > s0 > 0:
        per[int(xcc)].append((s0, s1, s2, int(l), b))
        cus[(int(xcc), (int(hwid) >> 8) & 0xff)].append((s0, s2, int(l), b))
CLK = 100e6   # s_memtime ticks at 100 MHz on this part? printed raw as well
for xcc, v in sorted(per.items()):
    t0 = min(a for a, *_ in v)
    end = max(e for _, _, e, _, _ in v) - t0
    print(f"XCC {xcc}: {len(v)} workgroups, span {end} ticks")
    v.sort()
    for a, m, e, l, b in v[:6] + v[len(v) // 2: len(v) // 2 + 4] + v[-8:]:
        print(f"   wg {b:5d} rows {l:3d} strips {(l + 31) // 32}  start {a - t0:8d}  loop {m - a:8d}  epilogue {e - m:6d}  end {e - t0:8d}")
    by = collections.defaultdict(list)
    for a, m, e, l, b in v:
        by[(l + 31) // 32].append(e - a)
    for ns, d in sorted(by.items()):
        d.sort()
        print(f"   strips {ns}: n {len(d):3d} life min {d[0]:7d} med {d[len(d) // 2]:7d} max {d[-1]:7d}")

print(f"{len(cus)} distinct (XCC, CU) pairs")
busy = []
for key, v in sorted(cus.items()):
    v.sort()
    t0 = v[0][0]
    life = sum(e - a for a, e, _, _ in v)
    busy.append((v[-1][1] - t0, life, len(v)))
    if key[0] == 0:
        print(f"XCC {key[0]} cu {key[1]:#04x}: " + "  ".join(f"[{a - t0:7d} +{e - a:6d} r{l}]" for a, e, l, _ in v))
busy.sort()
print("per CU: span of its workgroups (ticks) min / med / max:", busy[0][0], busy[len(busy) // 2][0], busy[-1][0], " units per CU min / max:", min(b[2] for b in busy), max(b[2] for b in busy))

st = (ctypes.c_ulonglong * (4 * 64))()
fn3 = L.ace_debug_dh_stages
fn3.restype = ctypes.c_int
fn3.argtypes = [ctypes.c_void_p]
if fn3(st) == 0:
    for w in range(4):
        v = [st[64 * w + k] for k in range(64)]
        if not v[0]:
            continue
        b_ = 264 * w
        print(f"traced workgroup {w} (block {b_}, rows {spans[6 * b_ + 3]}): kernel entry to the top of stage 0: {v[0] - spans[6 * b_]} ticks; stage = top | wait B | barrier | compute (ticks)")
        for t_ in range(12):
            a, b, c, d = v[4 * t_: 4 * t_ + 4]
            nxt = v[4 * t_ + 4] if t_ < 11 else v[48]
            print(f"   stage {t_:2d}: {b - a:6d} {c - b:6d} {d - c:6d} {nxt - d:6d}")
        print(f"   after the loop to the end: {v[49] - v[48]}")
