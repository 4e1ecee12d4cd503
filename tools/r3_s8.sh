#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q -k "healpix or hooks or sht or dhconv_op or physics" 2>&1 | grep -v "^  File\|Extension modules" | tail -25 > gpurun_out/s8_pytest.txt; tail -12 gpurun_out/s8_pytest.txt | cut -c1-200
timeout 200 python tools/bench_sht.py > gpurun_out/s8_bench_sht.json 2> gpurun_out/s8_bench_sht.err; cat gpurun_out/s8_bench_sht.json; tail -2 gpurun_out/s8_bench_sht.err
ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_wltrace.so timeout 300 python tools/trace_wl.py > gpurun_out/s8_trace_wl.txt 2>&1; tail -20 gpurun_out/s8_trace_wl.txt
