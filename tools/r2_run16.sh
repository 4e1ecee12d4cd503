#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/repeat_check.py 2000 2>&1 | tail -1
ACE_CONV_SPLIT=all python tools/repeat_check.py 1000 2>&1 | tail -1
ACE_CONV_SPLIT=none python tools/repeat_check.py 1000 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q 2>&1 | tail -3
bash tools/kdur2.sh dh2; grep "dhconv\|steps/s" gpurun_out/kdur_dh2.txt | head -3
