#!/usr/bin/env python
"""The reference's own SHT micro-benchmarks on the native transforms: `sht` and `inverse_sht` of
fme/sht_fix.py:232-310 = RealSHT(180, 360) / InverseRealSHT(180, 360) (default grid "lobatto": lmax 179, mmax 181) on
x = randn(1024, 180, 360), through the public module API (ace_amd.RealSHT: C ABI ace_sht_forward / ace_sht_inverse, including
the conversion between the internal spectral layout and the reference's (n, L, M) complex64).
usage: python tools/bench_sht.py [--n 1024] [--iters 10]  -> one JSON line"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n, H, W = args.n, 180, 360
x = torch.randn(n, H, W, device=dev)
out = {"workload": f"fme sht / inverse_sht benchmarks: RealSHT({H}, {W}) lobatto, batch {n}", "peak_GBps": 8000.0}
for prec in ("fp32", "f16x3"):
    f = ace_amd.RealSHT(H, W, precision=prec).to(dev)
    i = ace_amd.InverseRealSHT(H, W, precision=prec).to(dev)
    L, M = f.lmax, f.mmax
    alg = n * H * W * 4 + n * L * M * 8 + M * L * H * 4          # field + coefficients + Legendre table, each once
    c = f(x)
    y = i(c)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("sht", lambda: f(x)), ("inverse_sht", lambda: i(c))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        res[name] = {"us": round(us, 1), "GBps_algorithmic": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000.0, 4)}
    # round trip of a band-limited field (size-independent property): inverse(forward(.)) is the identity on its range
    y2 = i(f(y))
    res["roundtrip_rel_err"] = float((y2 - y).abs().max() / y.abs().max())
    res["algorithmic_MB"] = round(alg / 1e6, 1)
    out[prec] = res
print(json.dumps(out))
