#!/bin/bash
# round 4, fourth GPU call: persistent FFT kernels with next-unit prefetch, conditional layer norm for any field size, csfno goldens
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sht or reference_held or conditional_layer_norm or constant_field or quarter_degree_grid" 2>&1 | tail -25 > gpurun_out/r4_c4_new_tests.txt; tail -4 gpurun_out/r4_c4_new_tests.txt
bash tools/kdur2.sh c4_base $GRAFT_REPO_ROOT/exp/libexp_base.so; grep "steps/s" gpurun_out/kdur_c4_base.txt
bash tools/kdur2.sh c4_new; grep "steps/s" gpurun_out/kdur_c4_new.txt
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_ffttrace.so timeout 300 python tools/trace_fft.py > gpurun_out/r4_c4_fft_trace.txt 2>&1; tail -14 gpurun_out/r4_c4_fft_trace.txt
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_c4_pytest.txt; tail -3 gpurun_out/r4_c4_pytest.txt
