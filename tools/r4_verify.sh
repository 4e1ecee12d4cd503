#!/bin/bash
# round-4 verification of a binary: FFT source A/B, full GPU suite, default bench (with the CPU baseline), kernel trace + stats of a
# short bench, the four PMC passes (stamped with the library's sha256), noise-conditioned / HEALPix / SHT micro-benchmarks, 0.25 degree
# usage: gpurun -- bash tools/r4_verify.sh TAG [nopmc] [noquarter]
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out

bash tools/kdur2.sh ${tag}; grep "steps/s" gpurun_out/kdur_${tag}.txt
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt; tail -3 gpurun_out/${tag}_pytest.txt
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; head -c 400 gpurun_out/${tag}_bench.json; echo
if [ "$2" != "nopmc" ]; then bash tools/pmc_collect.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -25 gpurun_out/${tag}_pmc.log; fi
timeout 300 python tools/bench_csfno.py --no-oracle > gpurun_out/${tag}_bench_csfno.json 2> gpurun_out/${tag}_bench_csfno.err; head -c 600 gpurun_out/${tag}_bench_csfno.json; echo
timeout 200 python tools/bench_healpix.py > gpurun_out/${tag}_bench_healpix.json 2> gpurun_out/${tag}_bench_healpix.err; cat gpurun_out/${tag}_bench_healpix.json
timeout 200 python tools/bench_sht.py > gpurun_out/${tag}_bench_sht.json 2> gpurun_out/${tag}_bench_sht.err; cat gpurun_out/${tag}_bench_sht.json
if [ "$3" != "noquarter" ]; then timeout 500 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/${tag}_quarter.json 2> gpurun_out/${tag}_quarter.err; head -c 1500 gpurun_out/${tag}_quarter.json; tail -2 gpurun_out/${tag}_quarter.err; fi
