#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes over the NoiseConditionedSFNO at the ERA5 configuration (tools/bench_csfno.py's): two EAGER
forwards so that every kernel is dispatched individually.  usage: rocprofv3 --pmc <counters> --kernel-trace -f csv -d DIR -o NAME --
python tools/pmc_run_csfno.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402

IMG = (180, 360)
CFG = dict(embed_dim=512, noise_embed_dim=32, noise_type="gaussian", filter_type="linear", use_mlp=True, num_layers=8,
           operator_type="dhconv", affine_norms=True, normalize_big_skip=True)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(CFG)).build(44, 50, ace_amd.DatasetInfo(IMG)).torch_module.to(dev)
net.set_precision("f16x3")
x = torch.randn(1, 44, *IMG, device=dev)
noise = torch.randn(1, 32, *IMG, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net(x, noise=noise)
torch.cuda.synchronize()
print("pmc_run_csfno done", tuple(y.shape), bool(torch.isfinite(y).all()))
