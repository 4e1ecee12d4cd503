#!/bin/bash
# round 4, first GPU call: the never-run GPU tests + whole suite on the tree with all four prepared patches applied, the full-size
# property script, then a same-box per-kernel A/B of base / each patch alone / all four (tools/kdur2.sh), then a default bench.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ACE_RUN_UNVERIFIED=1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_insolation.py tests/test_healpix_resamplers.py tests/test_step_options.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r4_c1_new_tests.txt; tail -5 gpurun_out/r4_c1_new_tests.txt
timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_c1_pytest.txt; tail -3 gpurun_out/r4_c1_pytest.txt
for v in base all sel gelu leg m0; do
  if [ $v = all ]; then bash tools/kdur2.sh c1_$v; else bash tools/kdur2.sh c1_$v $GRAFT_REPO_ROOT/exp/libexp_$v.so; fi
  grep "steps/s" gpurun_out/kdur_c1_$v.txt
done
timeout 500 python tools/full_size_properties.py > gpurun_out/r4_c1_full_size.txt 2>&1; tail -8 gpurun_out/r4_c1_full_size.txt
timeout 300 python bench.py > gpurun_out/r4_c1_bench.json 2> gpurun_out/r4_c1_bench.err; head -c 300 gpurun_out/r4_c1_bench.json; echo
