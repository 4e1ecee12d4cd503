#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base fabl1 fabl2 fabl3 fabl4 frows8 frows32; do
  if [ $v = base ]; then bash tools/kdur2.sh f_$v; else bash tools/kdur2.sh f_$v $PWD/exp/libexp_$v.so; fi
  echo "== $v"; grep "dft_\|steps/s" gpurun_out/kdur_f_$v.txt
done
