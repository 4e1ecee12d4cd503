#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_collect.sh r02f > gpurun_out/pmc_collect_r02f.log 2>&1
bash tools/kdur2.sh final
python tools/repeat_check.py 2000 > gpurun_out/repeat_check_final.txt 2>&1; tail -1 gpurun_out/repeat_check_final.txt
for g in step window; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --graph $g > gpurun_out/bench_plain_$g.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --graph $g --hooks > gpurun_out/bench_hooks_$g.json 2> gpurun_out/bench_hooks_$g.err
done
python - <<PY
import json
for n in ("plain_step", "hooks_step", "plain_window", "hooks_window"):
    try:
        b = json.load(open(f"gpurun_out/bench_{n}.json")); print(n, b["value"], "steps/s", b["ms_per_step"], "ms")
    except Exception as e: print(n, "failed", e)
PY
timeout 900 python tools/bench_quarter_degree.py > gpurun_out/quarter_degree.json 2> gpurun_out/quarter_degree.err; tail -c 600 gpurun_out/quarter_degree.json
