#!/bin/bash
# round 4, second GPU call: folded Legendre stages (strip_fold.hip), forward-FFT trims, column-major staging of the inverse FFT, absmax
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sht or constant_field or full_size or quarter_degree_grid" 2>&1 | tail -15 > gpurun_out/r4_c2_sht_tests.txt; tail -4 gpurun_out/r4_c2_sht_tests.txt
bash tools/kdur2.sh c2_p4 $GRAFT_REPO_ROOT/exp/libexp_p4.so; grep "steps/s" gpurun_out/kdur_c2_p4.txt
bash tools/kdur2.sh c2_new; grep "steps/s" gpurun_out/kdur_c2_new.txt
ACE_NO_FOLD=1 bash tools/kdur2.sh c2_nofold; grep "steps/s" gpurun_out/kdur_c2_nofold.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r4_c2_pytest.txt; tail -3 gpurun_out/r4_c2_pytest.txt
