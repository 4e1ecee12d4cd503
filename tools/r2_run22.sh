#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -q -x -k "sht or transform or headline_network or quarter or roundtrip or dhconv_nets" 2>&1 | tail -3
bash tools/kdur2.sh fftnew
bash tools/kdur2.sh fftold $PWD/exp/libexp_fftold.so
grep "dft_\|steps/s" gpurun_out/kdur_fftnew.txt; grep "dft_\|steps/s" gpurun_out/kdur_fftold.txt
