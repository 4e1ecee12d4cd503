#!/bin/bash
# round 4, tenth GPU call: single-pass MFMA conditional layer norm (cln_mfma.hip) - parity tests, same-box A/B of the
# NoiseConditionedSFNO step at the ERA5 configuration (ACE_NO_CLN_MFMA=1 = the two-pass form), kernel durations of the new form
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conditional or noise_conditioned" 2>&1 | tail -6 > gpurun_out/r4_c10_tests.txt; tail -3 gpurun_out/r4_c10_tests.txt
for v in twopass mfma twopass2 mfma2; do
  case $v in twopass*) export ACE_NO_CLN_MFMA=1;; *) unset ACE_NO_CLN_MFMA;; esac
  timeout 400 python tools/bench_csfno.py --steps 20 --no-oracle > gpurun_out/r4_c10_csfno_$v.json 2> gpurun_out/r4_c10_csfno_$v.err
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r4_c10_csfno_{sys.argv[1]}.json"))
print(sys.argv[1], d.get("ms_per_step"), "ms/step", {k: round(v["ms_per_step"] * 1e3 / max(v["launches"], 1)) for k, v in d.get("stages", {}).items() if "norm" in k})
PY
done
unset ACE_NO_CLN_MFMA
timeout 400 python tools/bench_csfno.py --steps 3 > gpurun_out/r4_c10_csfno_parity.json 2> gpurun_out/r4_c10_csfno_parity.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c10_csfno_parity.json')); print(d.get('parity'))"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_c10 -o o -- python tools/bench_csfno.py --steps 5 --no-oracle > /dev/null 2>&1
f=$(ls /tmp/p_c10/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_c10_kernel_stats.csv && head -14 $f | cut -c1-150
