#!/bin/bash
# round 4, fourteenth GPU call: the HEALPix UNet forward captured in a hipGraph (eager vs replay, bit equality), its kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "healpix" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -15 > gpurun_out/r4_c14_tests.txt; tail -5 gpurun_out/r4_c14_tests.txt
timeout 300 python tools/bench_healpix.py --iters 30 > gpurun_out/r4_c14_healpix.json 2> gpurun_out/r4_c14_healpix.err; cat gpurun_out/r4_c14_healpix.json; tail -3 gpurun_out/r4_c14_healpix.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_c14 -o o -- python tools/bench_healpix.py --iters 10 > /dev/null 2>&1
f=$(find /tmp/p_c14 -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_c14_healpix_kernel_stats.csv && head -14 $f | cut -c1-170
exit 0
