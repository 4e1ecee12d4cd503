#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or dhconv_nets or packed or taps or graph_replay or config_variants" 2>&1 | tail -12 > gpurun_out/pytest_r2h.txt
tail -6 gpurun_out/pytest_r2h.txt
bash tools/kdur2.sh dh
ACE_NO_DHCONV_STRIP=1 bash tools/kdur2.sh nodh
head -14 gpurun_out/kdur_dh.txt; grep "steps/s\|dhconv" gpurun_out/kdur_nodh.txt
