#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in f11 f55 f46 f43; do
  bash tools/kdur2.sh $v $PWD/exp/libexp_$v.so
  echo "== $v"; grep "dft_forward\|steps/s" gpurun_out/kdur_$v.txt
done
ACE_SFNO_LIB=$PWD/exp/libexp_f46.so ACE_LIB=$PWD/exp/libexp_f46.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sht or dft or fft or quarter or transform" 2>&1 | tail -2
