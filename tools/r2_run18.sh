#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "headline_network or fused_mlp_shapes or dhconv_nets or packed or taps or graph_replay or config_variants or repeatab" 2>&1 | tail -5
bash tools/kdur2.sh dh3
ACE_NO_DHCONV_STRIP=1 bash tools/kdur2.sh nodh3
grep "dhconv\|steps/s" gpurun_out/kdur_dh3.txt | head -3;  grep "gemm4.*false, false\|steps/s" gpurun_out/kdur_nodh3.txt
python tools/repeat_check.py 1000 2>&1 | tail -1
