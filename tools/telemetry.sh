#!/bin/bash
# clocks and power of the GPU while the headline step runs back to back (rocm-smi samples beside a long bench run)
# usage: telemetry.sh TAG
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/telemetry_$1.txt
: > $out
echo "---- idle" >> $out
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" >> $out
echo "---- python bench.py --steps 6000 --warmup 20 (graph replays back to back); one sample per second, t = seconds since start" >> $out
t0=$(date +%s)
python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --precision f16x3 > /tmp/tele_bench.json 2>/tmp/tele_bench.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  t=$(( $(date +%s) - t0 ))
  line=$(rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use" | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';')
  echo "t=$t $line" >> $out
  sleep 1
done
wait $pid
python - >> $out <<'PY'
import json
b = json.load(open("/tmp/tele_bench.json"))
print("bench: %.1f steps/s, %.3f ms/step over %d steps" % (b["value"], b["ms_per_step"], b["steps"]))
PY
tail -50 $out
