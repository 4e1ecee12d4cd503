#!/usr/bin/env python
"""BASELINE configs[4]: the HEALPix UNet (reference default channel schedule 136 / 68 / 34, ConvNeXt blocks with capped GELU,
dilations 1 / 2 / 4, average pooling, transposed-convolution upsampling) at nside 64 (12 x 64 x 64 = 49 152 cells, ~1 degree),
44 in / 50 out channels, B = 1, random init: ms per forward on one MI355X (compensated-fp16 MFMA, one contraction per
convolution).  usage: python tools/bench_healpix.py [--nside 64] [--iters 10]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nside", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
cap = {"cap_value": 10}
cfg = dict(
    encoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=4, activation=cap),
                 down_sampling_block=dict(block_type="AvgPool", pooling=2), n_channels=[136, 68, 34], dilations=[1, 2, 4]),
    decoder=dict(conv_block=dict(block_type="ConvNeXtBlock", kernel_size=3, upscale_factor=4, activation=cap),
                 up_sampling_block=dict(block_type="TransposedConvUpsample", stride=2, activation=cap),
                 output_layer=dict(block_type="BasicConvBlock", kernel_size=1, n_layers=1), n_channels=[34, 68, 136], dilations=[4, 2, 1]),
    hpx_padding_mode="karlbauer")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ace_amd.ModuleSelector(type="HEALPixUNet", config=cfg).build(44, 50, ace_amd.DatasetInfo((args.nside, args.nside))).torch_module.to(dev)
x = torch.randn(1, 12, 44, args.nside, args.nside, device=dev)
flops = 0
for m in net.modules():      # dense conv flops at the resolution each layer runs at (bookkeeping only)
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        flops += 0           # resolution is not known here; reported as measured time only
with torch.no_grad():
    y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = net(x)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
out = {"workload": f"HEALPixUNet 136/68/34 ConvNeXt, nside {args.nside}, 44 -> 50 channels, B=1, f16x3 MFMA", "ms_per_forward": round(ms, 3),
       "forwards_per_s": round(1e3 / ms, 2), "finite": bool(torch.isfinite(y).all()), "parameters": sum(p.numel() for p in net.parameters())}
# the same forward captured in a hipGraph (ace_amd.CapturedHEALPixForward): device time without the Python-driven launches
try:
    cap_fwd = ace_amd.CapturedHEALPixForward(net, x)
    yg = cap_fwd(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        yg = cap_fwd(x)
    e1.record()
    torch.cuda.synchronize()
    out["ms_per_forward_captured"] = round(e0.elapsed_time(e1) / args.iters, 3)
    out["captured_equals_eager"] = bool(torch.equal(yg, y))
except Exception as exc:   # noqa: BLE001 - a capture failure is a result to report, the eager number above stands
    out["capture_error"] = f"{type(exc).__name__}: {exc}"[:300]
print(json.dumps(out))
