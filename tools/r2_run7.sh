#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_r2g.txt
tail -8 gpurun_out/pytest_r2g.txt
timeout 600 python bench.py --steps 40 --warmup 5 > gpurun_out/bench_r2g.json 2> gpurun_out/bench_r2g.err; tail -c 600 gpurun_out/bench_r2g.json
for g in step window; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --graph $g > gpurun_out/bench_plain_$g.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --graph $g --hooks > gpurun_out/bench_hooks_$g.json 2> gpurun_out/bench_hooks_$g.err
  python - <<PY
import json
for n in ("plain_$g", "hooks_$g"):
    try:
        b = json.load(open(f"gpurun_out/bench_{n}.json")); print(n, b["value"], "steps/s", b["ms_per_step"], "ms")
    except Exception as e: print(n, "failed", e)
PY
done
timeout 900 python tools/bench_quarter_degree.py > gpurun_out/quarter_degree.json 2> gpurun_out/quarter_degree.err; tail -c 1500 gpurun_out/quarter_degree.json; tail -3 gpurun_out/quarter_degree.err
