#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_mlp or headline_network or sht or packed or taps or dhconv_nets or graph_replay or race or config_variants or quarter" 2>&1 | tail -12 > gpurun_out/pytest_r2d.txt
tail -6 gpurun_out/pytest_r2d.txt
bash tools/kdur2.sh base4
bash tools/kdur2.sh fd1 $GRAFT_REPO_ROOT/exp/libexp_fd1.so
bash tools/kdur2.sh fd3 $GRAFT_REPO_ROOT/exp/libexp_fd3.so
grep "mlp_strip\|legendre_strip\|steps/s" gpurun_out/kdur_base4.txt gpurun_out/kdur_fd1.txt gpurun_out/kdur_fd3.txt
