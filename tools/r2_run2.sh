#!/bin/bash
# round-2 GPU session 2: fused MLP parity (headline shape), full suite, same-box A/B, PMC passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -s -k "fused_mlp or weight_update" 2>&1 | tail -25 > gpurun_out/pytest_r2b_mlp.txt
tail -8 gpurun_out/pytest_r2b_mlp.txt
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | tail -40 > gpurun_out/pytest_r2b.txt
tail -12 gpurun_out/pytest_r2b.txt
bash tools/kdur2.sh base2
ACE_NO_MLP_STRIP=1 bash tools/kdur2.sh nomlp
bash tools/pmc_collect.sh r02
