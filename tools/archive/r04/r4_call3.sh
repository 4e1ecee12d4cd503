#!/bin/bash
# round 4, third GPU call: encoder conv1 -> planes on the packed engine, side-stream weight packer, reference-held csfno goldens,
# full-size tests in the suite, FFT in-kernel timelines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_held or headline or graph_replay" 2>&1 | tail -15 > gpurun_out/r4_c3_new_tests.txt; tail -4 gpurun_out/r4_c3_new_tests.txt
bash tools/kdur2.sh c3_new; grep "steps/s" gpurun_out/kdur_c3_new.txt
ACE_NO_ENC_PK=1 ACE_NO_SIDE_PACK=1 bash tools/kdur2.sh c3_off; grep "steps/s" gpurun_out/kdur_c3_off.txt
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_ffttrace.so timeout 300 python tools/trace_fft.py > gpurun_out/r4_c3_fft_trace.txt 2>&1; tail -40 gpurun_out/r4_c3_fft_trace.txt
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_c3_pytest.txt; tail -3 gpurun_out/r4_c3_pytest.txt
