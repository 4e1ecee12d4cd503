#!/usr/bin/env python
"""Can the two kernels of the forward SHT overlap?  ("channel-chunked launches on two captured streams so the
Legendre of chunk i runs under the FFT of chunk i+1".)  Two independent forward transforms of 384 x 180 x 360 fields - the network's
own problem - each longitude FFT -> Legendre on its own plan and buffers:
  sequential: A then B on one stream;   concurrent: A on stream 1, B on stream 2 (so that FFT(B) can run under Legendre(A) etc.)
If the kernels complemented each other (one bound by the vector ALUs, one by the matrix cores) the concurrent time would approach
the longer of the two; if both already use the memory system the chip has, it stays at the sum.
usage: python tools/sht_overlap.py [--n 384] [--iters 20]  -> one JSON line"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=384)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n, H, W = args.n, 180, 360
fa = ace_amd.RealSHT(H, W, grid="legendre-gauss", precision="f16x3").to(dev)
fb = ace_amd.RealSHT(H, W, grid="legendre-gauss", precision="f16x3").to(dev)
xa, xb = torch.randn(n, H, W, device=dev), torch.randn(n, H, W, device=dev)
fa(xa); fb(xb)
torch.cuda.synchronize()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def sequential():
    for _ in range(args.iters):
        fa(xa)
        fb(xb)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(args.iters):
        with torch.cuda.stream(s1):
            fa(xa)
        with torch.cuda.stream(s2):
            fb(xb)
    cur.wait_stream(s1); cur.wait_stream(s2)


def single():
    for _ in range(args.iters):
        fa(xa)


t1 = timed(single) / args.iters
ts = timed(sequential) / args.iters
tc = timed(concurrent) / args.iters
print(json.dumps({"workload": f"two forward SHTs of {n} x {H} x {W} (f16x3; FFT + Legendre + API layout conversion each)",
                  "one_transform_us": round(t1, 1), "two_sequential_us": round(ts, 1), "two_concurrent_us": round(tc, 1),
                  "concurrent_over_sequential": round(tc / ts, 3)}))
