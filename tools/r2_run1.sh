#!/bin/bash
# round-2 GPU session 1: parity of the strip Legendre kernels, then same-box A/B of {strip, no strip, 32-row FFT, register epilogue}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_r2a.txt
cat gpurun_out/pytest_r2a.txt | tail -5
bash tools/kdur2.sh base
ACE_NO_STRIP=1 bash tools/kdur2.sh nostrip
bash tools/kdur2.sh fft32 $GRAFT_REPO_ROOT/exp/libexp_fft32.so
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_regepi.so timeout 600 python -m pytest tests -m gpu -x -q -k "packed or taps or modulus or dhconv_nets" 2>&1 | tail -5 > gpurun_out/pytest_regepi.txt
bash tools/kdur2.sh regepi $GRAFT_REPO_ROOT/exp/libexp_regepi.so
head -30 gpurun_out/kdur_base.txt
