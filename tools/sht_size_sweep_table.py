#!/usr/bin/env python
"""table of tools/sht_size_sweep.py runs: usage sht_size_sweep_table.py KERNEL_TRACE_CSV [COUNTER_CSV]"""
import collections
import csv
import sys

SIZES = (192, 384, 768, 1536, 3072)
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
       if "dft_forward" in r["Kernel_Name"] or "legendre" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for i, (name, us) in enumerate(seq):          # 4 forwards per size, two kernels each, in order
    per[(SIZES[i // 8], "fft" if "dft" in name else "legendre")].append(us)
fetch = collections.defaultdict(list)
if len(sys.argv) > 2:
    crow = [r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE"
            and ("dft_forward" in r["Kernel_Name"] or "legendre" in r["Kernel_Name"])]
    for i, r in enumerate(crow):
        fetch[(SIZES[i // 8], "fft" if "dft" in r["Kernel_Name"] else "legendre")].append(float(r["Counter_Value"]) * 1024 / 1e6)   # KB -> MB (x2 for 16-byte-per-lane reads: MICROARCH guide)
print("%6s %8s | %10s %12s | %10s %12s | %s" % ("n", "X MB", "fft us", "us / 384", "legendre", "us / 384", "FETCH_SIZE MB (fft, legendre; uncorrected)"))
for n in SIZES:
    a, b = sorted(per[(n, "fft")])[1:], sorted(per[(n, "legendre")])[1:]
    fa = min(a) if a else float("nan")
    fb = min(b) if b else float("nan")
    ft = "%.0f %.0f" % (min(fetch[(n, "fft")]), min(fetch[(n, "legendre")])) if fetch else "-"
    print("%6d %8.0f | %10.1f %12.1f | %10.1f %12.1f | %s" % (n, n * 180 * 181 * 8 / 1e6, fa, fa * 384 / n, fb, fb * 384 / n, ft))
