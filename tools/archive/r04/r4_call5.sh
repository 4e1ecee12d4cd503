#!/bin/bash
# round 4, fifth GPU call: persistent FFTs (second form: no young loads, no waterfalls, no scratch), big-K folded Legendre (0.25 deg),
# csfno block goldens with encoder_layers 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sht or reference_held or constant_field or quarter_degree" 2>&1 | tail -25 > gpurun_out/r4_c5_new_tests.txt; tail -4 gpurun_out/r4_c5_new_tests.txt
bash tools/kdur2.sh c5_base $GRAFT_REPO_ROOT/exp/libexp_base.so; grep "steps/s" gpurun_out/kdur_c5_base.txt
bash tools/kdur2.sh c5_new; grep "steps/s" gpurun_out/kdur_c5_new.txt
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_ffttrace.so timeout 300 python tools/trace_fft.py > gpurun_out/r4_c5_fft_trace.txt 2>&1; tail -14 gpurun_out/r4_c5_fft_trace.txt
timeout 400 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/r4_c5_quarter.json 2> gpurun_out/r4_c5_quarter.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4_c5_quarter.json"))
    print("quarter degree", d["ms_per_step"], "ms/step")
    for k, v in d["stages"].items(): print("  %-28s %8.1f us/launch" % (k, v["us_per_launch"]))
except Exception as e: print("quarter bench failed", e)
PY
tail -3 gpurun_out/r4_c5_quarter.err
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_c5_pytest.txt; tail -3 gpurun_out/r4_c5_pytest.txt
