#!/bin/bash
# round-3 GPU session 1: hazard reproducer, full GPU suite of the pruned / re-routed library, same-box A/B of the FFT and
# conv_ws compile-time variants (exp/libexp_{A..E}.so built by tools/mkvar.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./exp/store_hazard 128 > gpurun_out/s1_store_hazard.txt 2>&1; tail -4 gpurun_out/s1_store_hazard.txt )
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/s1_pytest.txt; tail -5 gpurun_out/s1_pytest.txt
bash tools/kdur2.sh s1_base
for v in A B C D E; do
  [ -f exp/libexp_$v.so ] && bash tools/kdur2.sh s1_$v $GRAFT_REPO_ROOT/exp/libexp_$v.so
done
bash tools/kdur2.sh s1_base2
for t in base A B C D E base2; do echo "== $t"; grep "dft_\|conv_ws\|steps/s" gpurun_out/kdur_s1_$t.txt | cut -c1-150; done
