#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/kdur2.sh splitA
ACE_NO_CONV_SPLIT=1 bash tools/kdur2.sh nosplitA
grep "conv_split\|steps/s" gpurun_out/kdur_splitA.txt; grep "conv_strip\|gemm4\|steps/s" gpurun_out/kdur_nosplitA.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "headline_network and f16x3 or fused_mlp_shapes" 2>&1 | tail -1; done
