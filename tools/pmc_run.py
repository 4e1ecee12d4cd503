#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes: a few EAGER forwards (no hipGraph) of the ACE2-shape network so that every hot
kernel is dispatched individually.  usage: rocprofv3 --pmc <counters> --kernel-trace -f csv -d DIR -o NAME -- python tools/pmc_run.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ace_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision(os.environ.get("ACE_SFNO_PRECISION", "f16x3"))
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev)
with torch.no_grad():
    for _ in range(reps + 1):
        y = net(x)
torch.cuda.synchronize()
print("pmc_run done", tuple(y.shape), float(y.abs().max()))
print("lib_sha256", bench.lib_sha256(), _lib.LIB_PATH)
from ace_amd import build as _build
print("src_sha256", _build.source_sha256())
