#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or taps or graph_replay or config_variants or weight_update or repeatab and ws" 2>&1 | tail -3
bash tools/kdur2.sh encws
ACE_NO_ENC_WS=1 bash tools/kdur2.sh noencws
grep "steps/s\|encoder\|norm0" gpurun_out/kdur_encws.txt; grep "steps/s\|encoder\|norm0" gpurun_out/kdur_noencws.txt
