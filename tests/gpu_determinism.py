"""eager-vs-eager and eager-vs-graph bitwise determinism probe (prints max abs differences)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import build_native_net
from oracle.sfno import SFNOConfig, init_state
dev = torch.device("cuda")
for (C, hw, L) in [(16, (24, 48), 2), (32, (45, 90), 3), (64, (24, 48), 2)]:
    cfg = SFNOConfig(in_chans=4, out_chans=4, img_shape=hw, embed_dim=C, num_layers=L, operator_type="dhconv")
    net = build_native_net(cfg, init_state(cfg, seed=6), dev)
    x = torch.randn(1, 4, *hw, device=dev)
    out = torch.empty(1, 4, *hw, device=dev)
    with torch.no_grad():
        a, ta = net.forward_with_taps(x)
        b, tb = net.forward_with_taps(x)
        print(C, hw, "eager vs eager:", (a - b).abs().max().item(), [float((p - q).abs().max()) for p, q in zip(ta, tb)])
        ref = net(x).clone()
        for i in range(3):
            net.forward_graph(x, out)
            torch.cuda.synchronize()
            print("   graph replay", i, "vs eager:", (out - ref).abs().max().item())
