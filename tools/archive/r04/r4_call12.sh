#!/bin/bash
# round 4, twelfth GPU call: conditional layer norm writes the packed convolutions' operand itself (P-format planes, bound-scaled):
# parity tests, same-box A/B of the NoiseConditionedSFNO step at the ERA5 configuration (ACE_NO_CLN_PLANES=1 = fp32 + pack pass),
# parity of one forward against the CPU oracle, kernel durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conditional or noise_conditioned" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -15 > gpurun_out/r4_c12_tests.txt; tail -4 gpurun_out/r4_c12_tests.txt
for v in fp32pack planes fp32pack2 planes2; do
  unset ACE_NO_CLN_PLANES
  case $v in fp32pack*) export ACE_NO_CLN_PLANES=1;; esac
  timeout 400 python tools/bench_csfno.py --steps 20 --no-oracle > gpurun_out/r4_c12_csfno_$v.json 2> gpurun_out/r4_c12_csfno_$v.err
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r4_c12_csfno_{sys.argv[1]}.json"))
print(sys.argv[1], d.get("ms_per_step"), "ms/step", {k: round(v["ms_per_step"] * 1e3 / max(v["launches"], 1)) for k, v in d.get("stages", {}).items()})
PY
done
unset ACE_NO_CLN_PLANES
timeout 400 python tools/bench_csfno.py --steps 3 > gpurun_out/r4_c12_csfno_parity.json 2> gpurun_out/r4_c12_csfno_parity.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c12_csfno_parity.json')); print(d.get('parity'))"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_c12 -o o -- python tools/bench_csfno.py --steps 5 --no-oracle > /dev/null 2>&1
f=$(find /tmp/p_c12 -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_c12_kernel_stats.csv && head -12 $f | cut -c1-160
exit 0
