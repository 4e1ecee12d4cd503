#!/usr/bin/env python
"""Micro-benchmarks of single kernels through the C ABI (for rocprofv3 --pmc passes and A/B builds).
usage: python tools/microbench.py [fc1|fc2|skip|sht|all] [--iters N]   (ACE_LIB=/path/to/variant.so to A/B)"""

import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ace_amd import _lib  # noqa: E402

if os.environ.get("ACE_LIB"):
    _lib.LIB_PATH = os.environ["ACE_LIB"]

dev = torch.device("cuda")


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def conv(cout, cin, hw, act, iters, name):
    L = _lib.lib()
    x = torch.randn(1, cin, hw, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.02
    b = torch.randn(cout, device=dev)
    y = torch.empty(1, cout, hw, device=dev)
    st = _lib.current_stream()
    t = timeit(lambda: _lib.check(L.ace_conv1x1(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, cin, cout, hw, act, st)), iters)
    fl = 2.0 * cout * cin * hw
    print(f"{name}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF", flush=True)


def sht(iters):
    import ace_amd
    n = 384
    x = torch.randn(n, 180, 360, device=dev)
    f = ace_amd.RealSHT(180, 360, 180, 181, "legendre-gauss")
    i = ace_amd.InverseRealSHT(180, 360, 180, 181, "legendre-gauss")
    c = f(x)
    t = timeit(lambda: f(x), iters)
    print(f"sht fwd (api, incl. layout conversion): {t * 1e6:8.1f} us", flush=True)
    t = timeit(lambda: i(c), iters)
    print(f"sht inv (api, incl. layout conversion): {t * 1e6:8.1f} us", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    print("lib:", _lib.LIB_PATH)
    if which in ("fc1", "all"):
        conv(768, 384, 64800, 1, iters, "fc1  768x384 GELU")
    if which in ("fc2", "all"):
        conv(384, 768, 64800, 0, iters, "fc2  384x768     ")
    if which in ("skip", "all"):
        conv(384, 384, 64800, 1, iters, "skip 384x384 GELU")
    if which in ("big", "all"):
        conv(4096, 4096, 4096, 0, max(iters // 4, 2), "sq   4096^3      ")
    if which in ("sht", "all"):
        sht(iters)


def conv16(cout, cin, hw, act, iters, name):
    L = _lib.lib()
    x = torch.randn(1, cin, hw, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.02
    b = torch.randn(cout, device=dev)
    y = torch.empty(1, cout, hw, device=dev)
    st = _lib.current_stream()
    for _ in range(iters):
        _lib.check(L.ace_conv1x1_f16x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, cin, cout, hw, act, st))
    torch.cuda.synchronize()
    print(name, "done (time it from the kernel trace)")


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1].startswith("h")):
    which = sys.argv[1]
    if which in ("hfc1", "hall"):
        conv16(768, 384, 64800, 1, 3, "f16x3 fc1")
    if which in ("hfc2", "hall"):
        conv16(384, 768, 64800, 0, 3, "f16x3 fc2")


def mlp16(cin, hid, cout, hw, iters):
    L = _lib.lib()
    g = lambda *s: torch.randn(*s, device=dev)
    x, w1, b1, w2, b2 = g(1, cin, hw), g(hid, cin) * 0.05, g(hid) * 0.1, g(cout, hid) * 0.05, g(cout) * 0.1
    y = torch.empty(1, cout, hw, device=dev)
    for _ in range(iters):
        _lib.check(L.ace_mlp_f16x3(*[_lib.ptr(v) for v in (x, w1, b1, w2, b2, y)], 1, cin, hid, cout, hw, 1, _lib.current_stream()))
    torch.cuda.synchronize()
    print("packed mlp done (time it from the kernel trace)")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pmlp":
    mlp16(384, 768, 384, 64800, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
