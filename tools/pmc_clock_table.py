#!/usr/bin/env python
"""Per hot kernel: effective clock, matrix-pipe busy fraction and where the waves' cycles go, from the merged counter file of
tools/pmc_collect.sh (per-launch durations of the SAME pass that counted the cycles).
  effective clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration in that pass    (MI355X_MICROARCH.md, DVFS give-back)
  MFMA busy       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
  issuing / issue-stalled / parked = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES (disjoint, sum ~ 1)
  wave lifetime   = 4 x SQ_WAVE_CYCLES / SQ_WAVES cycles, as a fraction of the kernel's GRBM cycles per XCD
usage: pmc_clock_table.py profiles/rNN_pmc_counters_merged.json"""
import json
import sys

pmc = json.load(open(sys.argv[1]))
print("# GRBM_GUI_ACTIVE covers the whole dispatch - a few microseconds of launch and drain outside the kernel's own start / end timestamps -")
print("# so `clock` is an UPPER bound (the part's maximum is 2.4 GHz: values above it show the size of that margin, ~15 % at 40 us, ~3 % at 140 us)")
print("# and `MFMA busy` (busy cycles / those cycles) a LOWER bound by the same margin.  tools/clock_probe.hip has the three clocks side by side.")
print("%-52s %8s %9s %9s | %8s %9s %8s | %9s" % ("kernel (grid threads)", "us (pass)", "clock GHz", "MFMA busy", "issuing", "iss-stall", "parked", "wave life"))
for k, v in pmc.items():
    if not isinstance(v, dict) or "GRBM_GUI_ACTIVE" not in v or "SQ_WAVE_CYCLES" not in v or "_us_sq1" not in v:
        continue
    if not any(t in k for t in ("conv_w", "dhconv", "legendre", "dft_", "gemm")):
        continue
    us = v["_us_sq1"]          # duration of the launch in the pass that counted GRBM_GUI_ACTIVE and the SQ counters
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    wc = v["SQ_WAVE_CYCLES"]
    print("%-52s %8.1f %9.2f %9.3f | %8.2f %9.2f %8.2f | %9.2f" % (
        k[:52], us, cyc / us / 1e3, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc),
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc,
        4.0 * wc / max(v.get("SQ_WAVES", 1), 1) / cyc))
