#!/bin/bash
# round-3 GPU session 3: does VALU work overlap with MFMA work on a SIMD?  + in-kernel timeline of conv_wl
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 ./exp/mfma_valu_overlap > gpurun_out/s3_mfma_valu_overlap.txt 2>&1; cat gpurun_out/s3_mfma_valu_overlap.txt
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_wltrace.so ACE_CONV_WL=1 timeout 300 python tools/trace_wl.py > gpurun_out/s3_trace_wl.txt 2>&1; tail -12 gpurun_out/s3_trace_wl.txt
