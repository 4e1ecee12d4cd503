#!/bin/bash
# round-3 GPU session 4: epilogue riding in the MFMA stream - conv_ws (fine-grained interleave) and conv_wl v2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "fused_mlp or repeatab or headline_network or block_taps" 2>&1 | tail -8 > gpurun_out/s4_pytest.txt; tail -4 gpurun_out/s4_pytest.txt
bash tools/kdur2.sh s4_base
[ -f exp/libexp_nofine.so ] && bash tools/kdur2.sh s4_nofine $GRAFT_REPO_ROOT/exp/libexp_nofine.so
ACE_CONV_WL=1 bash tools/kdur2.sh s4_wl
for t in base nofine wl; do echo "== $t"; grep "conv_w\|steps/s" gpurun_out/kdur_s4_$t.txt | cut -c1-150; done
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_wltrace.so ACE_CONV_WL=1 timeout 300 python tools/trace_wl.py > gpurun_out/s4_trace_wl.txt 2>&1; tail -12 gpurun_out/s4_trace_wl.txt
