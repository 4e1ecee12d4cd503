#!/usr/bin/env python
"""Prints a rocprofv3 --stats kernel table (o_kernel_stats.csv) compactly: usage kstats.py DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-100s n %5s avg %9.1f us  %5s%%" % (r["Name"].replace("void ace::", "").replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "")[:5]))
