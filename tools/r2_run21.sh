#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or packed or taps or graph_replay or config_variants or repeatab" 2>&1 | tail -6
bash tools/kdur2.sh wsall
ACE_CONV_WS=fc2 bash tools/kdur2.sh wsfc2
grep "conv_ws\|steps/s" gpurun_out/kdur_wsall.txt; grep "conv_ws\|conv_strip\|steps/s" gpurun_out/kdur_wsfc2.txt
