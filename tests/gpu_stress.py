"""Race screen: many repeated forwards (eager, library hipGraph, torch-captured window) must be bitwise identical."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import build_native_net
from oracle.sfno import SFNOConfig, init_state
dev = torch.device("cuda")
bad = 0
for (C, hw, L, B) in [(16, (24, 48), 2, 1), (32, (45, 90), 3, 2), (64, (24, 48), 2, 1), (48, (32, 64), 2, 3), (384, (180, 360), 2, 1)]:
    cfg = SFNOConfig(in_chans=4, out_chans=4, img_shape=hw, embed_dim=C, num_layers=L, operator_type="dhconv")
    net = build_native_net(cfg, init_state(cfg, seed=6), dev)
    x = torch.randn(B, 4, *hw, device=dev)
    out = torch.empty(B, 4, *hw, device=dev)
    with torch.no_grad():
        ref = net(x).clone()
        n_iter = 40 if C < 384 else 12
        mism = 0
        for i in range(n_iter):
            y = net(x)
            net.forward_graph(x, out)
            torch.cuda.synchronize()
            if not torch.equal(y, ref) or not torch.equal(out, ref):
                mism += 1
                print("  MISMATCH iter", i, float((y - ref).abs().max()), float((out - ref).abs().max()))
        # a second, busy stream to perturb timing
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            junk = torch.randn(4096, 4096, device=dev)
            for _ in range(10):
                junk = junk @ junk * 1e-4
        for i in range(n_iter):
            y = net(x)
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                mism += 1
                print("  MISMATCH (busy) iter", i, float((y - ref).abs().max()))
        print(f"C={C} hw={hw} L={L} B={B}: {mism} mismatches in {2 * n_iter} runs", flush=True)
        bad += mism
print("TOTAL MISMATCHES", bad)
