#!/bin/bash
# round-2 GPU session 3: deeper DMA rings (fused MLP, Legendre strip): parity subset + same-box A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_mlp or headline_network or sht or packed or taps or dhconv_nets or graph_replay or race" 2>&1 | tail -12 > gpurun_out/pytest_r2c.txt
tail -6 gpurun_out/pytest_r2c.txt
bash tools/kdur2.sh base3
ACE_NO_MLP_STRIP=1 bash tools/kdur2.sh nomlp3
grep "mlp_strip\|legendre_strip\|steps/s" gpurun_out/kdur_base3.txt gpurun_out/kdur_nomlp3.txt
