#!/usr/bin/env python
"""gpurun_out/pmc_<tag>.json (tools/pmc_collect.sh) -> profiles/r04_pmc_traffic.json keyed "mode|stage" for bench.py's
roofline.traffic, plus a readable table.  HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950: 16-byte-per-lane reads are counted
at half size, MI355X_MICROARCH.md) + WRITE_SIZE KB; kernels whose reads are 4-byte-per-lane get the uncorrected figure too."""
import json, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r04.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_pmc_traffic.json"
sha = sys.argv[3] if len(sys.argv) > 3 else None   # sha256 of the library the passes ran on (bench.py compares it)
src_sha = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else None   # ... and of its kernel sources + flags (ace_amd/build.py)
d = json.load(open(src))
RULES = [  # (regex on kernel key, stage, reads are wide (x2 correction applies))
    # dft_forward_fft_kernel<N1, N2, R, PLN, FULLM> (round 4 added FULLM)
    (r"^dft_forward_fft_kernel<\d+, \d+, \d+, true(, (true|false))?>", "forward_transform.dft", True),                 # planes input (every block but the first)
    (r"^dft_forward_fft_kernel<\d+, \d+, \d+, false(, (true|false))?>", "forward_transform.dft(first block)", True),   # fp32 rows
    (r"^dft_inverse_fft_kernel", "inverse_transform.dft", False),   # spectral side: 4-byte loads in 64-byte runs
    (r"^legendre_strip_kernel<0", "forward_transform.legendre", True),
    (r"^legendre_strip_kernel<1", "inverse_transform.legendre", True),
    # round 4: the folded kernels legendre_fold_kernel<OUT, FULLN, MODE> (strip_fold.hip); the strip is read with 4-byte loads
    (r"^legendre_fold_kernel<\d, (true|false), 0>", "forward_transform.legendre", False),
    (r"^legendre_fold_kernel<\d, (true|false), 1>", "inverse_transform.legendre", False),
    (r"^dhconv_strip_kernel", "dhconv", True),
    (r"^conv_ws_kernel<12, 1, 0>", "inner_skip+activation", True),
    (r"^conv_ws_kernel<12, 1, 1>", "mlp.fc1", True),
    (r"^conv_wl_kernel<24, 3", "mlp.fc1", True),
    (r"^conv_ws_kernel<12, 2, [24]>", "mlp.fc2+outer_skip", True),               # mode 4: residual from planes, h' as planes only
    (r"^conv_ws_kernel<12, 2, [35]>", "mlp.fc2+outer_skip(last block)", True),
    (r"^gemm3_f16x3_kernel<2, 2, false, true, false>", "decoder", True),
    (r"^gemm3_f16x3_kernel<2, 2, false, false, false>", "decoder", True),        # round 4: the decoder's first convolution only (the encoder's runs on gemm4)
    (r"^gemm4_f16x3_kernel<2, 2, false, true>", "encoder(first convolution)", True),
    (r"^conv_ws_kernel<12, 1, [28]>", "encoder", True),                            # round 4: last encoder convolution (the first one: gemm4 <2,2>)
]
out = {"_source": src, "_lib_sha256": sha, "_src_sha256": src_sha, "_note": "per-launch means over the dispatches of 2 eager forwards; separate rocprofv3 --pmc passes "
       "(FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS | SQ busy | SQ insts), never combined with API traces"}
rows = []
for k, v in d.items():
    for rx, stage, wide in RULES:
        if re.search(rx, k) and "FETCH_SIZE" in v:
            rd_raw = v["FETCH_SIZE"] * 1024
            rd = 2 * rd_raw if wide else rd_raw
            wr = v.get("WRITE_SIZE", 0.0) * 1024
            e = {"kernel": k, "bytes": int(rd + wr), "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1),
                 "read_MB_uncorrected": round(rd_raw / 1e6, 1), "wide_read_correction": wide,
                 "l2_hit_rate": v.get("l2_hit_rate"), "mfma_busy": v.get("mfma_busy"),
                 "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": v.get("SQ_LDS_IDX_ACTIVE"),
                 "dispatches": v.get("_dispatches")}
            out["f16x3|" + stage] = e
            rows.append((stage, e))
json.dump(out, open(dst, "w"), indent=1)
for stage, e in rows:
    print(f"{stage:34s} {e['kernel'][:44]:44s} read {e['read_MB']:7.1f} MB  write {e['write_MB']:7.1f} MB  L2 hit {e['l2_hit_rate']}  MFMA busy {e['mfma_busy']}")
