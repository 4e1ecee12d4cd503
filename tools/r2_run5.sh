#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_mlp or headline_network or packed or taps or graph_replay or race" 2>&1 | tail -12 > gpurun_out/pytest_r2e.txt
tail -6 gpurun_out/pytest_r2e.txt
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_trace.so timeout 300 python tools/trace_mlp.py 2>&1 | tail -34 > gpurun_out/trace_mlp2.txt
head -6 gpurun_out/trace_mlp2.txt; tail -3 gpurun_out/trace_mlp2.txt
bash tools/kdur2.sh base5
grep "mlp_strip\|legendre_strip\|steps/s" gpurun_out/kdur_base5.txt
