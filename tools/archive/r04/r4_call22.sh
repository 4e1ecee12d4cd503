#!/bin/bash
# round 4, call 22: block-diagonal (grouped) spectral filter - zero blocks skipped in dhconv_strip: goldens (groups 1 / 8), the reference's
# csfno_block benchmark shape with 1 and 8 groups, headline kernel duration check (dhconv must not move)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "noise_conditioned or dhconv" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -6 > gpurun_out/r4_c22_tests.txt; tail -3 gpurun_out/r4_c22_tests.txt
for g in 1 8; do timeout 300 python tools/bench_csfno_block.py --groups $g > gpurun_out/r4_c22_block_g$g.json 2> gpurun_out/r4_c22_block_g$g.err; python - $g <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r4_c22_block_g{sys.argv[1]}.json")); print("groups", sys.argv[1], d.get("block_ms"), d.get("block_stages_ms"))
PY
done
bash tools/kdur2.sh c22 > /dev/null 2>&1; grep "dhconv\|steps/s" gpurun_out/kdur_c22.txt | head -4
exit 0
