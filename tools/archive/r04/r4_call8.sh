#!/bin/bash
# round 4, eighth GPU call: inverse FFT with XCD-contiguous units where the spectral runs are shorter than a line (0.25 degree):
# same-box A/B of the 0.25-degree step, the SHT / FFT parity tests, the side-stream ensemble-mean test
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "sht or quarter or async or full_width" 2>&1 | tail -6 > gpurun_out/r4_c8_tests.txt; tail -3 gpurun_out/r4_c8_tests.txt
for v in prev new; do
  if [ $v = prev ]; then export ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_prev.so; else unset ACE_SFNO_LIB; fi
  timeout 400 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/r4_c8_quarter_$v.json 2> gpurun_out/r4_c8_quarter_$v.err
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r4_c8_quarter_{sys.argv[1]}.json"))
print(sys.argv[1], d["ms_per_step"], "ms/step", {k: round(v["us_per_launch"]) for k, v in d["stages"].items() if "dft" in k or "legendre" in k})
PY
done
