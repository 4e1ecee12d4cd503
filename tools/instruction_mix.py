#!/usr/bin/env python
"""Static instruction mix of the hot kernels (hipcc -S, device only, no GPU needed): per kernel the whole-kernel counts and every
basic block that holds MFMAs (the unrolled main loops).  valu excludes v_mfma; salu excludes s_waitcnt / s_nop / s_barrier.
usage: python tools/instruction_mix.py > profiles/rNN_instruction_mix.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ace_amd import build  # noqa: E402

HOT = {   # source -> substrings of the demangled kernel names to report
    "conv_ws.hip": ["conv_ws_kernel<12, 1, 0>", "conv_ws_kernel<12, 1, 2>", "conv_ws_kernel<12, 2, 4>", "conv_ws_kernel<12, 2, 5>"],
    "conv_wl.hip": ["conv_wl_kernel<24, 3, 4, true>", "conv_wl_kernel<32, 2, 4, false>"],
    "dhconv_strip.hip": ["dhconv_strip_kernel"],
    "strip_fold.hip": ["legendre_fold_kernel<0, true, 1>", "legendre_fold_kernel<1, true, 0>", "legendre_fold_big_kernel"],
    "fft.hip": ["dft_forward_fft_kernel<20, 18, 16, true, true>", "dft_inverse_fft_kernel<20, 18, 32, false>"],
    "cln_mfma.hip": ["cln_mfma_wide_kernel<16, true>", "cln_mfma_wide_kernel<8, true>", "cln_mfma_kernel<3>"],
    "kernels.hip": ["gemm4_f16x3_kernel<2, 2, true, false>", "gemm4_f16x3_kernel<2, 2, false, true>", "gemm4_implicit_kernel<2, 2, true>",
                    "gemm4_implicit_kernel<2, 2, false>", "gemm3_f16x3_kernel<2, 2, false, false, false>"],
}


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop") or op.startswith("s_endpgm") or op.startswith("s_code_end"):
        return None
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return None


KEYS = ["instr", "mfma", "valu", "lds", "vmem", "salu", "waitcnt", "barrier"]


def fmt(c):
    return "  ".join("%s %5d" % (k, c.get(k, 0)) for k in KEYS)


def main():
    print(f"# static instruction mix of the hot kernels (hipcc -S of the round-4 sources, kernel-source hash {build.source_sha256()[:12]}): whole kernel, then every")
    print("# basic block that holds MFMAs (the unrolled main loops).  valu excludes v_mfma; salu excludes s_waitcnt / s_nop / s_barrier.")
    for src, wanted in HOT.items():
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "o.s")
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                            os.path.join(ROOT, "ace_amd", "csrc", src)], check=True, capture_output=True)
            text = open(out).read()
        # kernels: "<mangled>:" ... "s_endpgm"
        names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        for mangled, name in zip(names, dem):
            name = name.replace("ace::(anonymous namespace)::", "").replace("ace::", "").replace("void ", "")
            if not any(w in name for w in wanted):
                continue
            m = re.search(r"^" + re.escape(mangled) + r":.*?s_endpgm", text, re.M | re.S)
            if not m:
                continue
            whole, blocks, cur, label = {}, [], {}, "entry"
            for line in m.group(0).splitlines()[1:]:
                line = line.split(";")[0].strip()
                if not line:
                    continue
                lm = re.match(r"^(\.LBB\w+):", line)
                if lm:
                    blocks.append((label, cur))
                    label, cur = lm.group(1), {}
                    continue
                if line.startswith("."):
                    continue
                cls = classify(line.split()[0])
                if cls is None:
                    continue
                for d in (whole, cur):
                    d["instr"] = d.get("instr", 0) + 1
                    d[cls] = d.get(cls, 0) + 1
            blocks.append((label, cur))
            print()
            print(re.sub(r"\(.*", "", name)[:110])
            print("  whole kernel      " + fmt(whole))
            for lab, c in blocks:
                if c.get("mfma", 0) >= 6:
                    print("  block %-11s " % lab + fmt(c) + "   valu/mfma %.1f" % (c.get("valu", 0) / c["mfma"]))


if __name__ == "__main__":
    main()
