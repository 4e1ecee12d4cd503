#!/bin/bash
# Verification of the HEAD binary on one box: per-kernel durations (rocprofv3 --kernel-trace --stats), the whole GPU suite, the four PMC
# passes (stamped with the library's and the sources' sha256; installed as profiles/rNN_pmc_traffic.json ON THE BOX so that the bench line
# taken right after carries traffic / mfma_busy_pmc of THIS binary), the default bench (with the CPU baseline), the noise-conditioned /
# HEALPix / SHT micro-benchmarks, 0.25 degree.   usage: gpurun -- bash tools/verify.sh r06 [nopmc] [noquarter]
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/kdur2.sh ${tag}; grep "steps/s" gpurun_out/kdur_${tag}.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt; tail -3 gpurun_out/${tag}_pytest.txt
if [ "$2" != "nopmc" ]; then
  bash tools/pmc_collect.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1; tail -14 gpurun_out/pmc_${tag}_table.txt
  cp gpurun_out/pmc_${tag}_traffic.json profiles/${tag}_pmc_traffic.json
  python tools/pmc_clock_table.py gpurun_out/pmc_${tag}.json > gpurun_out/${tag}_pmc_clock_busy_waits.txt 2>/dev/null
fi
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; head -c 1800 gpurun_out/${tag}_bench.json; echo
timeout 300 python tools/bench_csfno.py --no-oracle > gpurun_out/${tag}_bench_csfno.json 2> gpurun_out/${tag}_bench_csfno.err; head -c 300 gpurun_out/${tag}_bench_csfno.json; echo
timeout 200 python tools/bench_healpix.py > gpurun_out/${tag}_bench_healpix.json 2> gpurun_out/${tag}_bench_healpix.err; cat gpurun_out/${tag}_bench_healpix.json
timeout 200 python tools/bench_sht.py > gpurun_out/${tag}_bench_sht.json 2> gpurun_out/${tag}_bench_sht.err; head -c 400 gpurun_out/${tag}_bench_sht.json; echo
if [ "$3" != "noquarter" ]; then timeout 500 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/${tag}_quarter.json 2> gpurun_out/${tag}_quarter.err; head -c 300 gpurun_out/${tag}_quarter.json; echo; tail -2 gpurun_out/${tag}_quarter.err; fi
