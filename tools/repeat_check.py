#!/usr/bin/env python
"""N forwards of the headline network on one input: how many differ bitwise from the first, and by how much.
usage: [ACE_* switches] python tools/repeat_check.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision(os.environ.get("PREC", "f16x3"))
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
with torch.no_grad():
    y0 = net(x).clone()
    bad = 0; worst = 0.0; nbad_el = 0
    for _ in range(N):
        y = net(x)
        if not torch.equal(y, y0):
            bad += 1
            d = (y - y0).abs()
            worst = max(worst, float(d.max()))
            nbad_el = max(nbad_el, int((d > 0).sum()))
print(f"switches {[k + '=' + v for k, v in os.environ.items() if k.startswith('ACE_') and 'LIB' not in k]}: {bad} of {N} differ, "
      f"max |diff| {worst:.3e} (|y| max {float(y0.abs().max()):.3f}), most differing elements in one run {nbad_el}")
