#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then bash tools/kdur2.sh abl_base; else bash tools/kdur2.sh abl_$v $GRAFT_REPO_ROOT/exp/libexp_$v.so; fi
  echo "== $v"; grep "mlp_strip\|steps/s" gpurun_out/kdur_abl_$v.txt
done
