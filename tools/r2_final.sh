#!/bin/bash
# final evidence of the round: full GPU suite, PMC passes, kernel trace, repeatability, default bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/pytest_final.txt; tail -3 gpurun_out/pytest_final.txt
bash tools/pmc_collect.sh r02f > gpurun_out/pmc_collect_r02f.log 2>&1; grep "pass " gpurun_out/pmc_collect_r02f.log
bash tools/kdur2.sh final; head -12 gpurun_out/kdur_final.txt; grep "steps/s" gpurun_out/kdur_final.txt
python tools/repeat_check.py 2000 > gpurun_out/repeat_check_final.txt 2>&1; tail -1 gpurun_out/repeat_check_final.txt
( time timeout 600 python bench.py > gpurun_out/bench_final_default.json 2> gpurun_out/bench_final_default.err ) 2>&1 | grep real
head -c 1800 gpurun_out/bench_final_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/bench_csfno.py --no-oracle > gpurun_out/bench_csfno_r02.json 2> gpurun_out/bench_csfno_r02.err; head -c 400 gpurun_out/bench_csfno_r02.json
