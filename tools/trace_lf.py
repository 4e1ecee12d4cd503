#!/usr/bin/env python
"""Per-workgroup lifetimes and per-CU timelines of strip_fold.hip's legendre_fold_kernel (s_memtime stamps of wave 0).  Needs a library
built with -DACE_LF_TRACE: tools/mkvar.sh lftrace -DACE_LF_TRACE; ACE_SFNO_LIB=exp/libexp_lftrace.so python tools/trace_lf.py"""
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
bench.ACE2 = dict(bench.ACE2, num_layers=1)      # ONE block: one forward and one inverse Legendre launch per forward
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision("f16x3")
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net(x)
torch.cuda.synchronize()
L = _lib.lib()
N = 4096
spans = (ctypes.c_ulonglong * (2 * N * 5))()
fn = L.ace_debug_lf_spans
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(spans) == 0
for mode, name in ((0, "forward"), (1, "inverse")):
    cus = collections.defaultdict(list)
    life = collections.defaultdict(list)
    for b in range(N):
        s0, s1, m, hwid, xcc = (spans[(mode * N + b) * 5 + k] for k in range(5))
        if s1 > s0 > 0:
            cus[(int(xcc), (int(hwid) >> 8) & 0xff)].append((s0, s1, int(m), b))
            life[int(m) // 20].append(s1 - s0)
    print(f"== {name}: {sum(len(v) for v in cus.values())} workgroups on {len(cus)} CUs")
    for k, d in sorted(life.items()):
        d.sort()
        print(f"   m {20 * k:3d}..{20 * k + 19:3d}: n {len(d):3d} life min {d[0]:6d} med {d[len(d) // 2]:6d} max {d[-1]:6d}")
    rows = []
    for key, v in sorted(cus.items()):
        v.sort()
        t0 = v[0][0]
        rows.append((v[-1][1] - t0, sum(e - a for a, e, _, _ in v), len(v), max(e for _, e, _, _ in v) - t0))
        if key[0] == 0 and key[1] in (0x00, 0x01, 0x20, 0x21, 0x40, 0x61):
            print(f"   XCC 0 cu {key[1]:#04x}: " + " ".join(f"[{a - t0:6d}+{e - a:5d} m{m}]" for a, e, m, _ in v))
    rows.sort()
    print("   per CU: last end (ticks) min / med / max:", rows[0][3], rows[len(rows) // 2][3], rows[-1][3],
          " busy sum / 2 min / med / max:", min(r[1] for r in rows) // 2, sorted(r[1] for r in rows)[len(rows) // 2] // 2, max(r[1] for r in rows) // 2,
          " units per CU:", min(r[2] for r in rows), max(r[2] for r in rows))
