#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k healpix 2>&1 | tail -15 > gpurun_out/s7_pytest.txt; cat gpurun_out/s7_pytest.txt
python tools/bench_healpix.py > gpurun_out/s7_healpix.json 2> gpurun_out/s7_healpix.err; cat gpurun_out/s7_healpix.json; tail -3 gpurun_out/s7_healpix.err
ACE_SFNO_LIB=exp/libexp_g128.so ACE_LIB=exp/libexp_g128.so python tools/bench_healpix.py > gpurun_out/s7_healpix_g128.json 2> gpurun_out/s7_healpix_g128.err; cat gpurun_out/s7_healpix_g128.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_hpx -o o -- python tools/bench_healpix.py --iters 5 > /dev/null 2>&1
f=$(find /tmp/p_hpx -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/s7_healpix_kernel_stats.csv && head -12 $f | cut -c1-200
f=$(find /tmp/p_hpx -name o_kernel_trace.csv | head -1); [ -n "$f" ] && python - $f <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# one forward: list gemm3 launches with grid and duration of the last forward
g=[(r["Kernel_Name"][:40], r["Grid_Size_X"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "gemm3" in r["Kernel_Name"]]
n=len(g)//6
for k in g[-n:]: print(k)
PY
