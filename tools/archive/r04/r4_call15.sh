#!/bin/bash
# round 4, fifteenth GPU call: HEALPix k x k convolutions on the packed engine (implicit-GEMM gemm4) - parity tests, same-box A/B
# (ACE_HPX_NO_PACKED=1 = the fp32-operand engine), the forward captured in a hipGraph, kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "healpix or hpx" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -15 > gpurun_out/r4_c15_tests.txt; tail -5 gpurun_out/r4_c15_tests.txt
for v in unpacked packed unpacked2 packed2; do
  unset ACE_HPX_NO_PACKED
  case $v in unpacked*) export ACE_HPX_NO_PACKED=1;; esac
  timeout 300 python tools/bench_healpix.py --iters 30 > gpurun_out/r4_c15_healpix_$v.json 2> gpurun_out/r4_c15_healpix_$v.err; echo $v; cut -c95-400 gpurun_out/r4_c15_healpix_$v.json; tail -2 gpurun_out/r4_c15_healpix_$v.err | cut -c1-300
done
unset ACE_HPX_NO_PACKED
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_c15 -o o -- python tools/bench_healpix.py --iters 10 > /dev/null 2>&1
f=$(find /tmp/p_c15 -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_c15_healpix_kernel_stats.csv && head -8 $f | cut -c1-170
exit 0
