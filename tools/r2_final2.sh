#!/bin/bash
# last verification of the committed state: full GPU suite + default bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/pytest_final.txt; tail -3 gpurun_out/pytest_final.txt
timeout 200 python bench.py > gpurun_out/bench_final_default.json 2> gpurun_out/bench_final_default.err; head -c 600 gpurun_out/bench_final_default.json
