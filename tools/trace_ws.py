#!/usr/bin/env python
"""In-kernel timeline of conv_ws.hip mode 4 (fc2 + outer skip, K = 768): s_memtime stamps of wave 0 of one workgroup.  Needs a library
built with -DACE_X_TRACE=<workgroup>: tools/mkvar.sh wstrace -DACE_X_TRACE=0; ACE_SFNO_LIB=exp/libexp_wstrace.so python tools/trace_ws.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ace_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
bench.ACE2 = dict(bench.ACE2, num_layers=2)      # two blocks: the first one's fc2 runs mode 4 (planes out), once per forward
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
fn = L.ace_debug_trace_ws
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = list(buf)
print("tile: epilogue of the previous tile | store retire | pieces + residual issue | 36 MFMAs (1152 cycles of matrix pipe) | wait + barrier || stage 1: issue | 36 MFMAs | wait + barrier")
for pt in range(1, 12):
    e = t[16 * pt: 16 * pt + 16]
    if not e[0] or not t[16 * pt + 16]:
        break
    print(f"  tile {pt:2d}: {e[1] - e[0]:6d} {e[2] - e[1]:6d} {e[3] - e[2]:6d} {e[4] - e[3]:6d} {e[5] - e[4]:6d} || {e[11] - e[8]:6d} {e[12] - e[11]:6d} {e[13] - e[12]:6d}   total {t[16 * pt + 16] - e[0]:6d}")
