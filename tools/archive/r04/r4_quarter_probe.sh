#!/bin/bash
# 0.25-degree step, where the time goes and why: per-kernel durations of the real-shape network with the
# planes-only residual stream on / off, and the HBM bytes + L2 hit rate of its kernels (is fc2's 22 % per-pixel excess over the
# 1-degree run the hidden tensor no longer fitting the 256 MB infinity cache?).
# usage: gpurun --timeout 1200 -- bash tools/r4_quarter_probe.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for stream in 1 0; do
  ACE_PLANES_STREAM=$stream timeout 300 python tools/bench_quarter_degree.py --steps 5 > gpurun_out/r4_quarter_stream$stream.json 2> gpurun_out/r4_quarter_stream$stream.err
  python - gpurun_out/r4_quarter_stream$stream.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["ms_per_step"], "ms/step")
for k, v in d["stages"].items():
    print("  %-28s %8.1f us/launch %s" % (k, v["us_per_launch"], v.get("algorithmic_GBps", "")))
PY
done
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/q_$tag
  timeout 420 rocprofv3 --pmc $grp --kernel-trace -f csv -d /tmp/q_$tag -o o -- python tools/bench_quarter_degree.py --steps 1 > /tmp/q_$tag.log 2>&1
  f=$(find /tmp/q_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py "$f" > gpurun_out/r4_quarter_pmc_$tag.json && head -c 1500 gpurun_out/r4_quarter_pmc_$tag.json
done
