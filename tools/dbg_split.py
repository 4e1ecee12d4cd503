#!/usr/bin/env python
"""debug: taps of the headline net with the current ACE_NO_CONV_SPLIT setting -> /tmp/taps_<tag>.pt; with 'cmp a b' prints where they differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
if sys.argv[1] == "cmp":
    a = torch.load(f"/tmp/taps_{sys.argv[2]}.pt"); b = torch.load(f"/tmp/taps_{sys.argv[3]}.pt")
    for i, (x, y) in enumerate(zip(a, b)):
        d = (x - y).abs()
        print(f"tap {i}: max diff {d.max().item():.3e}  ref max {x.abs().max().item():.3e}  nan {torch.isnan(y).sum().item()}")
        if d.max() > 1e-3 * x.abs().max():
            dc = d.amax(dim=(0, 2, 3)); bad = (dc > 1e-3 * x.abs().max()).nonzero().flatten()
            print("   bad channels:", bad[:40].tolist(), "count", bad.numel())
            dp = d.amax(dim=(0, 1)).flatten(); badp = (dp > 1e-3 * x.abs().max()).nonzero().flatten()
            print("   bad pixels: count", badp.numel(), "first", badp[:16].tolist(), "last", badp[-8:].tolist())
            if badp.numel():
                q = badp % 128
                print("   bad pixel mod 128 histogram (by 32):", [int(((q >= 32 * k) & (q < 32 * k + 32)).sum()) for k in range(4)])
            break
    sys.exit(0)
import bench
dev = torch.device("cuda", 0)
stepper, forcing, prog, diag = bench.build_stepper(dev, seed=0)
net = stepper.modules[0]
net.set_precision("f16x3")
x = torch.randn(1, len(forcing) + len(prog), *bench.IMG, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
with torch.no_grad():
    y, taps = net.forward_with_taps(x)
torch.save([t.cpu() for t in taps[:3]] + [y.cpu()], f"/tmp/taps_{sys.argv[1]}.pt")
print("saved", sys.argv[1])
