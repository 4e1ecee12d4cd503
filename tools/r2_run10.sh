#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or dhconv_nets or packed or taps or graph_replay or config_variants" 2>&1 | tail -25 > gpurun_out/pytest_r2i.txt
tail -12 gpurun_out/pytest_r2i.txt
bash tools/kdur2.sh split
ACE_NO_CONV_SPLIT=1 bash tools/kdur2.sh nosplit
head -12 gpurun_out/kdur_split.txt; grep "steps/s" gpurun_out/kdur_split.txt; grep "steps/s\|conv_strip\|gemm4" gpurun_out/kdur_nosplit.txt
