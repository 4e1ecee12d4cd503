#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -q -x -k "dhconv or headline_network or fused_mlp_shapes and ws or taps or config_variants" 2>&1 | tail -3
bash tools/kdur2.sh dhnew
bash tools/kdur2.sh dhold $PWD/exp/libexp_dhold.so
grep "dhconv\|steps/s" gpurun_out/kdur_dhnew.txt | head -2; grep "dhconv\|steps/s" gpurun_out/kdur_dhold.txt | head -2
