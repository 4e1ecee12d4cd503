#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or dhconv_nets or packed or taps or graph_replay or config_variants or determin" 2>&1 | tail -25 > gpurun_out/pytest_r2k.txt
tail -5 gpurun_out/pytest_r2k.txt
bash tools/kdur2.sh split3
head -8 gpurun_out/kdur_split3.txt; grep "steps/s" gpurun_out/kdur_split3.txt
