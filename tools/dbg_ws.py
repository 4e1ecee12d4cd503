#!/usr/bin/env python
"""debug: dump / compare internal workspaces of a small net (needs a -DACE_DEBUG_WS build in ACE_SFNO_LIB)
usage: dbg_ws.py run TAG C H W   |   dbg_ws.py cmp TAGA TAGB C H W"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
cmd = sys.argv[1]
if cmd == "cmp":
    ta, tb, C, H, W = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    HW = H * W
    for name, rows in (("U", 2 * C), ("P", C), ("P2", C)):
        a = np.load(f"/tmp/ws_{ta}_{name}.npy"); b = np.load(f"/tmp/ws_{tb}_{name}.npy")
        nb = a.size // (2 * rows * HW)
        a = a[: 2 * nb * rows * HW].reshape(2, nb, rows // 8, HW, 8).astype(np.float32)
        b = b[: 2 * nb * rows * HW].reshape(2, nb, rows // 8, HW, 8).astype(np.float32)
        a = a[0:1] + a[1:2]; b = b[0:1] + b[1:2]   # value = hi + lo
        d = np.abs(a - b); d[np.isnan(d)] = np.inf
        print(name, "nan in a", int(np.isnan(a).sum()), "nan in b", int(np.isnan(b).sum()))
        print(name, "values max diff", d.max(), "ref max", np.abs(a).max())
        if d.max() > 1e-3 * np.abs(a).max():
            bad = np.argwhere(d > 1e-3 * np.abs(a).max())
            print("  count", len(bad), "of", d.size)
            for dim, nm in enumerate(["plane(hi/lo)", "sample", "kgroup", "pixel", "e"]):
                vals, cnt = np.unique(bad[:, dim], return_counts=True)
                print("  ", nm, "distinct", len(vals), "first", vals[:24].tolist(), "counts", cnt[:8].tolist())
            px = np.unique(bad[:, 3]); print("   pixel % 32 set:", sorted(set((px % 32).tolist())))
            kg = np.unique(bad[:, 2]); print("   kgroup % 4 set:", sorted(set((kg % 4).tolist())))
            k = bad[0]; print("   first bad", k.tolist(), "a", a[tuple(k)], "b", b[tuple(k)])
            wg = np.unique(bad[:, 3] // 128); print("   pixel // 128 (workgroups):", wg[:20].tolist(), "count", len(wg))
            for kk in bad[:12]: print("     ", kk.tolist(), a[tuple(kk)], b[tuple(kk)])
            # does b match a at another location?  search for b's bad value in a over the same pixel
            kk = bad[0]; col = a[0, kk[1], :, kk[3], :].reshape(-1); hit = np.argwhere(np.abs(col - b[tuple(kk)]) < 1e-3)
            print("   b's first bad value found in a at rows (same pixel):", hit.flatten()[:8].tolist(), "expected row", int(kk[2]) * 8 + int(kk[4]))
    a = np.load(f"/tmp/ws_{ta}_part.npy").view(np.float32); b = np.load(f"/tmp/ws_{tb}_part.npy").view(np.float32)
    print("part sizes", a.size, b.size, "nan a", np.isnan(a).sum(), "nan b", np.isnan(b).sum())
    nb = np.argwhere(np.isnan(b)).flatten()
    if nb.size:
        e4 = nb // 4; comp = nb % 4
        print("  nan float4 entries: first", e4[:12].tolist(), "last", e4[-6:].tolist(), "comp set", sorted(set(comp.tolist())))
        print("  entry // C:", sorted(set((e4 // C).tolist()))[:40], " entry % C range", (e4 % C).min(), (e4 % C).max())
        d = np.abs(a - b); d[np.isnan(d)] = 0
        bad = np.argwhere(d > 1e-3 * np.abs(a).max()).flatten()
        print("  non-nan mismatches:", bad.size, "first", (bad[:12] // 4).tolist())
    ya = np.load(f"/tmp/ws_{ta}_y.npy"); yb = np.load(f"/tmp/ws_{tb}_y.npy")
    print("y max diff", np.abs(ya - yb).max(), "nan", np.isnan(yb).sum())
    sys.exit(0)
tag, C, H, W = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
from oracle.sfno import SFNOConfig, init_state
from _util import build_native_net
from ace_amd import _lib
dev = torch.device("cuda", 0)
cfg = SFNOConfig(in_chans=6, out_chans=5, img_shape=(H, W), embed_dim=C, num_layers=int(os.environ.get("DBG_LAYERS", "1")), operator_type="dhconv")
state = init_state(cfg, seed=17)
x = torch.randn(1, 6, H, W, generator=torch.Generator().manual_seed(18)) * 0.8 + 0.5
net = build_native_net(cfg, state, dev, "f16x3")
with torch.no_grad():
    y = net(x.to(dev))
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.ace_debug_read_workspace.restype = ctypes.c_long
raw.ace_debug_read_workspace.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
for name in ("U", "P", "P2", "part"):
    buf = np.zeros(1 << 27, dtype=np.float16)
    got = raw.ace_debug_read_workspace(net._native, name.encode(), buf.ctypes.data, buf.nbytes)
    print(name, "bytes", got)
    if got > 0:
        np.save(f"/tmp/ws_{tag}_{name}.npy", buf[: got // 2])
np.save(f"/tmp/ws_{tag}_y.npy", y.cpu().numpy())
