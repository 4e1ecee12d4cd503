#!/usr/bin/env python
"""Does it matter to the forward SHT's two kernels whether the intermediate X (the longitude spectrum, written by the FFT kernel and
read by the Legendre kernel) is still on the die?    RealSHT(180, 360, legendre-gauss, f16x3: the network's
kernels) on n fields for n = 192 ... 3072: X is 50 MB at n = 192 (fits the 256 MB Infinity Cache several times over), 100 MB at
the network's n = 384, 800 MB at n = 3072 (cannot be on the die when the Legendre kernel reads it).  Run under
`rocprofv3 --kernel-trace` (durations per launch) and again under `--pmc FETCH_SIZE` (HBM bytes); tools/sht_size_sweep_table.py
makes the table.  If the per-field time of the two kernels does not change from 384 to 3072, the X round trip through HBM is not
what bounds them and an L2 / MALL hand-off has nothing to win."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402

dev = torch.device("cuda", 0)
H, W = 180, 360
f = ace_amd.RealSHT(H, W, H, W // 2 + 1, "legendre-gauss", precision="f16x3").to(dev)
for n in (192, 384, 768, 1536, 3072):
    x = torch.randn(n, H, W, device=dev)
    for _ in range(4):
        c = f(x)
    torch.cuda.synchronize()
    print("n", n, "X MB", n * H * (W // 2 + 1) * 8 / 1e6, flush=True)
    del x, c
