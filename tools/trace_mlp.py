#!/usr/bin/env python
"""In-kernel timeline of the fused MLP (s_memtime stamps of wave 0 of one workgroup); needs a library built with
-DACE_X_TRACE=<block id> (tools/mkvar.sh trace -DACE_X_TRACE=300).  usage: ACE_SFNO_LIB=exp/libexp_trace.so python tools/trace_mlp.py"""
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
assert raw.ace_debug_mlp_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
t0 = int(buf[0])
r = lambda i: int(buf[i]) - t0
print("start->loads issued", r(1), "loads landed", r(2), "after first barrier", r(3))
for q in list(range(0, 12)) + list(range(44, 52)) + list(range(88, 96)):
    b = 8 + 4 * q
    nxt = int(buf[8 + 4 * (q + 1)]) - t0 if q + 1 < 96 else r(4)
    print(f"group {q:3d} (chunk {q // 4} phase {q % 4}): top {r(b):8d}  vmcnt wait {r(b + 1) - r(b):6d}  barrier {r(b + 2) - r(b + 1):6d}  "
          f"dma issue {r(b + 3) - r(b + 2):6d}  compute {nxt - r(b + 3):6d}")
print("loop end", r(4), "tail drain", r(5) - r(4), "epilogue", r(6) - r(5), "total", r(6))
