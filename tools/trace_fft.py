#!/usr/bin/env python
"""In-kernel timeline of the longitude FFT kernels (csrc/fft.hip): s_memtime stamps of wave 0 of EVERY workgroup (phases of the
forward and of the inverse kernel) at the network's size (384 fields, 180 x 360).  The stamps live in the measurement version of
fft.hip (tools/patches/r4_persistent_fft_kernels.patch: persistent kernels + trace macros; the shipped fft.hip has neither):
  git apply tools/patches/r4_persistent_fft_kernels.patch; tools/mkvar.sh ffttrace -DACE_FFT_TRACE; git checkout ace_amd/csrc/fft.hip
  ACE_SFNO_LIB=exp/libexp_ffttrace.so python tools/trace_fft.py
Prints per-phase medians (cycles of the stamp counter), workgroup lifetimes, and how many workgroups were alive at once per XCC."""
import collections
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ace_amd  # noqa: E402
from ace_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
C, H, W = 384, 180, 360
fwd = ace_amd.RealSHT(H, W, H, W // 2 + 1, "legendre-gauss", precision="f16x3")
inv = ace_amd.InverseRealSHT(H, W, H, W // 2 + 1, "legendre-gauss", precision="f16x3")
x = torch.randn(C, H, W, device=dev)
L = _lib.lib()
fn = L.ace_debug_fft_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
NW = 8192
buf = (ctypes.c_ulonglong * (NW * 8))()


def report(title, names):
    """stamps are per UNIT (channel block, latitude) of the persistent kernels: 0 = the unit's turn starts (its loads were issued one
    unit earlier), then one stamp per phase; word 7 = XCC id << 32 | workgroup"""
    assert fn(buf, 0) == 0
    rows = [[buf[w * 8 + e] for e in range(8)] for w in range(NW)]
    rows = [r for r in rows if r[0] > 0]
    print(f"== {title}: {len(rows)} units")
    last = max(i for i in range(7) if rows[0][i] > 0)
    for i in range(last):
        d = sorted(r[i + 1] - r[i] for r in rows)
        print(f"  {names[i]:52s} med {d[len(d) // 2]:8d}  p10 {d[len(d) // 10]:8d}  p90 {d[9 * len(d) // 10]:8d}")
    life = sorted(r[last] - r[0] for r in rows)
    print(f"  {'unit turn':52s} med {life[len(life) // 2]:8d}  p10 {life[len(life) // 10]:8d}  p90 {life[9 * len(life) // 10]:8d}")
    per = collections.defaultdict(list)
    for r in rows:
        per[(r[7] >> 32, r[7] & 0xffffffff)].append((r[0], r[last]))
    spans = sorted(max(e for _, e in v) - min(a for a, _ in v) for v in per.values())
    units = sorted(len(v) for v in per.values())
    print(f"  {len(per)} workgroups, units each {units[0]}..{units[-1]}, lifetime med {spans[len(spans) // 2]} max {spans[-1]} cycles")


with torch.no_grad():
    for _ in range(3):
        c = fwd(x)
        y = inv(c)
    torch.cuda.synchronize()
    assert fn(None, 1) == 0
    c = fwd(x)
    torch.cuda.synchronize()
    report("forward FFT (rows in, spectral runs out)", ["wait for the prefetched rows, stage to LDS (+barrier)", "issue next unit's loads, level 1 (+2 barriers)", "level 2 (18-point FFTs) + store issue"])
    assert fn(None, 1) == 0
    y = inv(c)
    torch.cuda.synchronize()
    report("inverse FFT (spectral runs in, rows out)", ["wait for the prefetched entries, 18-point FFTs (+barrier)", "issue next unit's loads, step B (+2 barriers)", "copy-out issue"])
