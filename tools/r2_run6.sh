#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_mlp or headline_network or packed or taps or graph_replay or race" 2>&1 | tail -12 > gpurun_out/pytest_r2f.txt
tail -6 gpurun_out/pytest_r2f.txt
ACE_MLP_FUSED=1 timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_mlp or headline_network or packed or taps" 2>&1 | tail -12 > gpurun_out/pytest_r2f_split.txt
tail -4 gpurun_out/pytest_r2f_split.txt
bash tools/kdur2.sh base6
ACE_MLP_FUSED=1 bash tools/kdur2.sh split6
ACE_MLP_FUSED=1 ACE_NO_CONV_STRIP=1 bash tools/kdur2.sh old6
for t in base6 split6 old6; do echo == $t; grep "conv_strip\|mlp_strip\|gemm4\|steps/s\|inner_skip\|mlp.fc" gpurun_out/kdur_$t.txt; done
