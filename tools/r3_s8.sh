#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conditioned or csfno or noise" 2>&1 | tail -8 > gpurun_out/s8_pytest.txt; cat gpurun_out/s8_pytest.txt
python tools/bench_csfno.py --no-oracle > gpurun_out/s8_csfno.json 2> gpurun_out/s8_csfno.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/s8_csfno.json'))
print(b['ms_per_step'])
for k,v in b['stages'].items(): print(k, v['ms_per_step'])
PY
tail -3 gpurun_out/s8_csfno.err
