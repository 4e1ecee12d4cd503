#!/bin/bash
# First GPU call of the next round: everything that was written after round 3's GPU budget was spent and has only run on the CPU
# emulations (tests/_fake_sfno.py, tests/_fake_hpx.py) or not at all - the new GPU-marked tests first and on their own, then the
# whole suite, the full-size property script, a default bench.
# usage: gpurun --timeout 1500 -- bash tools/r4_first.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ACE_RUN_UNVERIFIED=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_insolation.py tests/test_healpix_resamplers.py tests/test_step_options.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_first_new_tests.txt; tail -5 gpurun_out/r4_first_new_tests.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4_first_pytest.txt; tail -3 gpurun_out/r4_first_pytest.txt
timeout 600 python tools/full_size_properties.py > gpurun_out/r4_first_full_size.txt 2>&1; cat gpurun_out/r4_first_full_size.txt | tail -8
timeout 400 python bench.py > gpurun_out/r4_first_bench.json 2> gpurun_out/r4_first_bench.err; head -c 300 gpurun_out/r4_first_bench.json; echo
