#!/usr/bin/env python
"""In-kernel timeline of conv_wl.hip (s_memtime stamps of wave 0 of one workgroup).  Needs a library built with
-DACE_X_TRACE=<workgroup>: tools/mkvar.sh wltrace -DACE_X_TRACE=0; ACE_SFNO_LIB=exp/libexp_wltrace.so ACE_CONV_WL=1 python tools/trace_wl.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ace_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
bench.ACE2 = dict(bench.ACE2, num_layers=1)      # ONE block: one conv_wl launch per forward, so the stamps are of one launch
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision("f16x3")
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net(x)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 512)()
fn = L.ace_debug_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = list(buf)
print(f"prologue (slots, weights -> LDS, barrier): {t[1] - t[0]} cycles")
k = 2
n = 0
while k + 2 < 512 and t[k + 1] > t[k] > 0:
    nxt = t[k + 2] if t[k + 2] > 0 else t[k + 1]
    print(f"tile {n}: MFMA stream with the previous tile's epilogue (216 MFMAs = 6912 cycles of matrix pipe) {t[k + 1] - t[k]:7d}   to the next stamp {nxt - t[k + 1]:6d}")
    k += 2
    n += 1
if n:
    print(f"{n} tiles, {max(t) - t[0]} cycles from kernel start to the end of the last epilogue")

# every workgroup's first / last stamp (wave 0): how long each lives, how late it starts (per XCC: the counters of different
# XCCs are not synchronised)
spans = (ctypes.c_ulonglong * (4096 * 3))()
fn2 = L.ace_debug_wg_spans
fn2.restype = ctypes.c_int
fn2.argtypes = [ctypes.c_void_p]
if fn2(spans) == 0:
    import collections
    per = collections.defaultdict(list)
    for b in range(4096):
        s0, s1, xcc = spans[3 * b], spans[3 * b + 1], spans[3 * b + 2]
        if s1 > s0 > 0:
            per[int(xcc)].append((s0, s1, b))
    for xcc, v in sorted(per.items()):
        t0 = min(a for a, _, _ in v)
        dur = sorted(e - a for a, e, _ in v)
        late = sorted(a - t0 for a, _, _ in v)
        end = max(e for _, e, _ in v) - t0
        print(f"XCC {xcc}: {len(v)} workgroups, lifetime min/med/max {dur[0]} / {dur[len(dur) // 2]} / {dur[-1]} cycles, "
              f"start delay med/max {late[len(late) // 2]} / {late[-1]}, last end {end} cycles after the first start")
