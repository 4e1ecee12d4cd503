#!/usr/bin/env python
"""In-kernel timeline of conv_split.hip (s_memtime stamps of wave 0 of one workgroup); needs a library built with
-DACE_X_TRACE=<block id> [-DACE_X_TRACE_MODE=<0..3>] (tools/mkvar.sh trace -DACE_X_TRACE=300 -DACE_X_TRACE_MODE=1).
usage: ACE_SFNO_LIB=exp/libexp_trace.so python tools/trace_split.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ace_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision("f16x3")
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net(x)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(512, dtype=np.uint64)
assert raw.ace_debug_split_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
t0 = int(buf[0])
r = lambda i: int(buf[i]) - t0
print("start -> scales", r(1), "-> before wait", r(2), "-> loads landed", r(3), "-> after barrier", r(4))
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for u in range(nst):
    b = 8 + 6 * u
    if b + 6 >= 512 or buf[b] == 0:
        break
    nxt = (int(buf[b + 6]) - t0) if (u + 1 < nst and b + 12 < 512 and buf[b + 6]) else r(5)
    print(f"stage {u:3d}: top {r(b):8d}  dma issue {r(b + 1) - r(b):6d}  pre-mfma {r(b + 2) - r(b + 1):6d}  steps {r(b + 3) - r(b + 2):6d}  "
          f"exchange {r(b + 4) - r(b + 3):6d}  vmcnt wait {r(b + 5) - r(b + 4):6d}  barrier+copy {nxt - r(b + 5):6d}  total {nxt - r(b):6d}")
print("loop end", r(5), "last epilogue", r(6) - r(5), "total", r(6))
