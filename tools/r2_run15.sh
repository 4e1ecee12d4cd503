#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_r2m.txt
tail -8 gpurun_out/pytest_r2m.txt
( time timeout 600 python bench.py > gpurun_out/bench_r2m.json 2> gpurun_out/bench_r2m.err ) 2>&1 | grep real
tail -c 1500 gpurun_out/bench_r2m.json
