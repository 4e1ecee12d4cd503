#!/usr/bin/env python
"""BASELINE configs[3] / SURVEY 8(d) "S2": the 0.25-degree SFNO (721 x 1440, L = M = 721) at the REAL shape - embed 384,
8 layers, 44 in / 50 out channels, B = 1 - for 10 forward steps on one MI355X: ms/step and per-stage times with their
algorithmic bytes / flops.  usage: python tools/bench_quarter_degree.py [--steps 10] [--embed 384] [--layers 8]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--embed", type=int, default=384)
ap.add_argument("--layers", type=int, default=8)
args = ap.parse_args()

from types import SimpleNamespace  # noqa: E402

from ace_amd import _lib  # noqa: E402
from ace_amd.sfno import SphericalFourierNeuralOperatorNet  # noqa: E402

H, W, C, NL = 721, 1440, args.embed, args.layers
dev = torch.device("cuda", 0)
t0 = time.time()
params = SimpleNamespace(operator_type="dhconv", scale_factor=1, embed_dim=C, num_layers=NL, data_grid="legendre-gauss")
torch.manual_seed(0)
net = SphericalFourierNeuralOperatorNet(params=params, in_chans=44, out_chans=50, img_shape=(H, W)).to(dev).eval()
net.set_precision("f16x3")
x = torch.randn(1, 44, H, W, device=dev)
y = torch.empty(1, 50, H, W, device=dev)
with torch.no_grad():
    net(x)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    net.forward_graph(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        net.forward_graph(x, y)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
L = _lib.lib()
ns = L.ace_sfno_num_stages()
tm = (ctypes.c_float * ns)()
calls = (ctypes.c_int * ns)()
_lib.check(L.ace_sfno_forward_timed(net._native, x.data_ptr(), y.data_ptr(), 1, _lib.current_stream(), tm, calls))
act, coef = C * H * W * 4, C * H * (W // 2 + 1) * 8
model = {"forward_transform.dft": act + coef, "inverse_transform.dft": act + coef,
         "forward_transform.legendre": 2 * coef + (W // 2 + 1) * H * H * 4, "inverse_transform.legendre": 2 * coef + (W // 2 + 1) * H * H * 4,
         "dhconv": 2 * coef + 2 * C * C * H * 4, "inner_skip+activation": 3 * act, "mlp.fc1": 3 * act, "mlp.fc2+outer_skip": 5 * act}
flops = {"forward_transform.legendre": 4 * C * (W // 2 + 1) * H * H, "inverse_transform.legendre": 4 * C * (W // 2 + 1) * H * H,
         "dhconv": 8 * C * C * H * (W // 2 + 1), "inner_skip+activation": 2 * C * C * H * W, "mlp.fc1": 4 * C * C * H * W,
         "mlp.fc2+outer_skip": 4 * C * C * H * W}
stages = {}
for i in range(ns):
    nm = L.ace_sfno_stage_name(i).decode()
    per = tm[i] / max(calls[i], 1)
    stages[nm] = {"ms_per_step": round(tm[i], 3), "launches": calls[i], "us_per_launch": round(per * 1e3, 1)}
    if nm in model and per > 0:
        stages[nm]["algorithmic_GBps"] = round(model[nm] / per / 1e6, 1)
    if nm in flops and per > 0:
        stages[nm]["dense_TFLOPs"] = round(flops[nm] / per / 1e9, 1)
print(json.dumps({"config": f"0.25 degree SFNO {H}x{W}, lmax=mmax={H}, embed {C}, {NL} layers, 44/50 channels, B=1, f16x3, random init",
                  "steps": args.steps, "ms_per_step": round(ms, 2), "steps_per_s": round(1e3 / ms, 3),
                  "simulated_years_per_day": round(1e3 / ms * 86400 / 1460, 1), "build_seconds": round(t_build, 1),
                  "finite": bool(torch.isfinite(y).all()), "hbm_peak_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                  "stages": stages}))
