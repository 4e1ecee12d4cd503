#!/usr/bin/env python
"""The reference's own `csfno_block` benchmark shape (fme/core/models/conditional_sfno/benchmark.py:27-46): ONE
FourierNeuralOperatorBlock of the conditional SFNO at B = 2, C = 512, 180 x 360, 64 noise channels, 3 labels, 32 positional
context channels, filter groups G.  The native library runs whole networks, so this times a 1-block NoiseConditionedSFNO of that
width (44 -> 50 channels around it) with the per-stage HIP-event timer and reports the BLOCK's stages (conditional norms,
SHT, spectral filter, inverse SHT, inner skip, MLP) apart from the encoder / decoder around them.
usage: python tools/bench_csfno_block.py [--groups 1] [--iters 5]  -> one JSON line"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402
from ace_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, C, IMG = 2, 512, (180, 360)
cfg = dict(embed_dim=C, noise_embed_dim=64, noise_type="gaussian", num_layers=1, use_mlp=True, pos_embed=False,
           context_pos_embed_dim=32, filter_num_groups=args.groups)


class Info:
    img_shape = IMG
    all_labels = {"a", "b", "c"}


torch.manual_seed(0)
net = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=cfg, conditional=True).build(44, 50, Info()).torch_module.to(dev)
g = torch.Generator().manual_seed(1)
with torch.no_grad():      # the conditioning weights start at zero in the reference: make them matter
    for k, p in net.named_parameters():
        if ".W_scale_" in k or ".W_bias_" in k:
            p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
x = torch.randn(B, 44, *IMG, generator=g).to(dev)
labels = torch.randn(B, 3, generator=g).to(dev)       # the reference benchmark draws its label embedding with randn
with torch.no_grad():
    y = net(x, labels=labels)
    torch.cuda.synchronize()
    # the merged conditioning field of that call, rebuilt for the timed entry point
    noise = net.draw_noise(B, dev)
    pos = net.pos_embed.detach().repeat(B, 1, 1, 1) + torch.einsum("bl,lpxy->bpxy", labels, net.label_pos_embed.detach())
    cond = torch.cat([noise, pos, labels[:, :, None, None].expand(B, 3, *IMG), torch.ones(B, 1, *IMG, device=dev)], dim=1).contiguous()
    L = _lib.lib()
    ns = L.ace_sfno_num_stages()
    ms = (ctypes.c_float * ns)()
    calls = (ctypes.c_int * ns)()
    out = torch.empty(B, 50, *IMG, device=dev)
    acc = [0.0] * ns
    for it in range(args.iters + 1):
        _lib.check(L.ace_sfno_forward_conditioned_timed(net._native, x.data_ptr(), cond.data_ptr(), out.data_ptr(), B,
                                                        _lib.current_stream(), ms, calls))
        if it > 0:
            for i in range(ns):
                acc[i] += ms[i] / args.iters
stages = {L.ace_sfno_stage_name(i).decode(): round(acc[i], 4) for i in range(ns)}
block = {k: v for k, v in stages.items() if k not in ("encoder", "decoder")}
print(json.dumps({"workload": f"conditional SFNO block, reference benchmark shape: B={B}, C={C}, {IMG[0]}x{IMG[1]}, noise 64, labels 3, "
                              f"pos 32, groups {args.groups}, f16x3", "block_ms": round(sum(block.values()), 3), "block_stages_ms": block,
                  "encoder_decoder_ms": round(stages["encoder"] + stages["decoder"], 3), "finite": bool(torch.isfinite(y).all())}))
