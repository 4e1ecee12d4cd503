#!/bin/bash
# round-4 verification of a binary: per-kernel durations (rocprofv3 --kernel-trace --stats), full GPU suite, the four PMC passes
# (stamped with the library's sha256; installed as profiles/r04_pmc_traffic.json on the box so that the bench line taken right after
# carries traffic / mfma_busy_pmc of THIS binary), default bench (with the CPU baseline), noise-conditioned / HEALPix / SHT
# micro-benchmarks, 0.25 degree.   usage: gpurun -- bash tools/r4_verify.sh TAG [nopmc] [noquarter]
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/kdur2.sh ${tag}; grep "steps/s" gpurun_out/kdur_${tag}.txt
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt; tail -3 gpurun_out/${tag}_pytest.txt
if [ "$2" != "nopmc" ]; then
  bash tools/pmc_collect.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -14 gpurun_out/pmc_${tag}_table.txt
  cp gpurun_out/pmc_${tag}_traffic.json profiles/r04_pmc_traffic.json
fi
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; head -c 1800 gpurun_out/${tag}_bench.json; echo
timeout 300 python tools/bench_csfno.py --no-oracle > gpurun_out/${tag}_bench_csfno.json 2> gpurun_out/${tag}_bench_csfno.err; head -c 300 gpurun_out/${tag}_bench_csfno.json; echo
timeout 200 python tools/bench_healpix.py > gpurun_out/${tag}_bench_healpix.json 2> gpurun_out/${tag}_bench_healpix.err; cat gpurun_out/${tag}_bench_healpix.json
timeout 200 python tools/bench_sht.py > gpurun_out/${tag}_bench_sht.json 2> gpurun_out/${tag}_bench_sht.err; head -c 400 gpurun_out/${tag}_bench_sht.json; echo
if [ "$3" != "noquarter" ]; then timeout 500 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/${tag}_quarter.json 2> gpurun_out/${tag}_quarter.err; head -c 300 gpurun_out/${tag}_quarter.json; echo; tail -2 gpurun_out/${tag}_quarter.err; fi
