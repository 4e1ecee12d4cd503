#!/bin/bash
# round 4, eleventh GPU call: conditional layer norm single pass, 128-pixel (16-byte) form against the 32-pixel form and the two-pass
# kernels - parity tests, same-box A/B of the NoiseConditionedSFNO step at the ERA5 configuration, kernel durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conditional or noise_conditioned" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -15 > gpurun_out/r4_c11_tests.txt; tail -4 gpurun_out/r4_c11_tests.txt
for v in twopass narrow wide twopass2 narrow2 wide2; do
  unset ACE_NO_CLN_MFMA ACE_CLN_NARROW
  case $v in twopass*) export ACE_NO_CLN_MFMA=1;; narrow*) export ACE_CLN_NARROW=1;; esac
  timeout 400 python tools/bench_csfno.py --steps 20 --no-oracle > gpurun_out/r4_c11_csfno_$v.json 2> gpurun_out/r4_c11_csfno_$v.err
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r4_c11_csfno_{sys.argv[1]}.json"))
print(sys.argv[1], d.get("ms_per_step"), "ms/step", {k: round(v["ms_per_step"] * 1e3 / max(v["launches"], 1)) for k, v in d.get("stages", {}).items() if "norm" in k})
PY
done
unset ACE_NO_CLN_MFMA ACE_CLN_NARROW
timeout 400 python tools/bench_csfno.py --steps 3 > gpurun_out/r4_c11_csfno_parity.json 2> gpurun_out/r4_c11_csfno_parity.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c11_csfno_parity.json')); print(d.get('parity'))"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_c11 -o o -- python tools/bench_csfno.py --steps 5 --no-oracle > /dev/null 2>&1
f=$(find /tmp/p_c11 -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_c11_kernel_stats.csv && head -16 $f | cut -c1-160
exit 0
