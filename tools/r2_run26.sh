#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or taps or graph_replay or config_variants or repeatab and ws" 2>&1 | tail -3
bash tools/kdur2.sh lstat
bash tools/kdur2.sh lstatold $PWD/exp/libexp_rstat.so
grep "steps/s\|conv_ws_kernel<12, 2, 2>\|conv_ws_kernel<12, 1, 2>\|instnorm_finalize" gpurun_out/kdur_lstat.txt; grep "steps/s\|conv_ws_kernel<12, 2, 2>\|conv_ws_kernel<12, 1, 2>\|instnorm_finalize" gpurun_out/kdur_lstatold.txt
