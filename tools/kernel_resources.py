#!/usr/bin/env python
"""Register / scratch / LDS table of every kernel in ace_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage, device-only,
no GPU needed).  usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ace_amd import build  # noqa: E402

FIELDS = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("TotalSGPRs", "SGPR"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "LDS")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    csrc = os.path.join(ROOT, "ace_amd", "csrc")
    print(f"# kernel resource usage (hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage); kernel-source hash {build.source_sha256()[:12]}")
    print("# %-13s %5s %5s %5s %8s %4s %7s  %s" % ("file", *[f[1] for f in FIELDS], "kernel"))
    with_scratch = []
    for src in build.SOURCES:
        if not src.endswith(".hip") or src == "capi.hip":
            continue
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "o.s"), os.path.join(csrc, src)],
                               capture_output=True, text=True)
        rows, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:[^ ]*:\d+:\d+: )?\s*(Function Name|[A-Za-z ]+(?:\[[^\]]+\])?): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2)
            if k == "Function Name":
                cur = {"name": v}
                rows.append(cur)
            elif cur is not None:
                cur[k] = v
        names = demangle([r_["name"] for r_ in rows])
        for r_, nm in zip(rows, names):
            nm = nm.replace("ace::(anonymous namespace)::", "").replace("ace::", "").replace("void ", "")
            nm = re.sub(r"\(.*", "", nm)[:100]
            vals = [r_.get(f[0], "?") for f in FIELDS]
            print("%-15s %5s %5s %5s %8s %4s %7s  %s" % (src[:-4], *vals, nm))
            if vals[3] not in ("0", "?"):
                with_scratch.append((src, nm, vals[3]))
    print("# kernels with scratch: " + ("none" if not with_scratch else "; ".join(f"{s}:{n} ({b} B/lane)" for s, n, b in with_scratch)))


if __name__ == "__main__":
    main()
