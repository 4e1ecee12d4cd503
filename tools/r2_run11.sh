#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or dhconv_nets or packed or taps or graph_replay or config_variants" 2>&1 | tail -25 > gpurun_out/pytest_r2j.txt
tail -5 gpurun_out/pytest_r2j.txt
ACE_SFNO_LIB=$PWD/exp/libexp_trace1.so ACE_LIB=$PWD/exp/libexp_trace1.so timeout 300 python tools/trace_split.py 24 > gpurun_out/trace_split_fc1.txt 2>&1
ACE_SFNO_LIB=$PWD/exp/libexp_trace2.so ACE_LIB=$PWD/exp/libexp_trace2.so timeout 300 python tools/trace_split.py 24 > gpurun_out/trace_split_fc2.txt 2>&1
cat gpurun_out/trace_split_fc1.txt | tail -30; tail -28 gpurun_out/trace_split_fc2.txt
