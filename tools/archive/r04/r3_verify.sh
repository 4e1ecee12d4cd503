#!/bin/bash
# round-3 verification of a binary: full GPU suite, default bench (with the CPU baseline), kernel trace of a short bench,
# the four PMC passes (stamped with the library's sha256), the reference's SHT micro-benchmarks, the 0.25-degree bench
# usage: gpurun -- bash tools/r3_verify.sh TAG [nopmc] [noquarter]
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt; tail -3 gpurun_out/${tag}_pytest.txt
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; head -c 400 gpurun_out/${tag}_bench.json; echo
bash tools/kdur2.sh ${tag}; grep "steps/s" gpurun_out/kdur_${tag}.txt
timeout 200 python tools/bench_healpix.py > gpurun_out/${tag}_bench_healpix.json 2> gpurun_out/${tag}_bench_healpix.err; cat gpurun_out/${tag}_bench_healpix.json
timeout 200 python tools/bench_sht.py > gpurun_out/${tag}_bench_sht.json 2> gpurun_out/${tag}_bench_sht.err; cat gpurun_out/${tag}_bench_sht.json
if [ "$2" != "nopmc" ]; then bash tools/pmc_collect.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -25 gpurun_out/${tag}_pmc.log; fi
if [ "$3" != "noquarter" ]; then timeout 500 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/${tag}_quarter.json 2> gpurun_out/${tag}_quarter.err; head -c 1500 gpurun_out/${tag}_quarter.json; tail -2 gpurun_out/${tag}_quarter.err; fi
