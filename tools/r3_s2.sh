#!/bin/bash
# round-3 GPU session 2: hazard probe with the software-mechanism variant, fused physics + conv_wl + shrinking-batch tests,
# conv_wl A/B against conv_ws, hooks overhead with the fused physics
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./exp/store_hazard 64 > gpurun_out/s2_store_hazard.txt 2>&1; tail -6 gpurun_out/s2_store_hazard.txt )
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -q -x -k "physics or fused_mlp or repeatab or shrinking or checkpoint or window_graph or windowed or dhconv_at" 2>&1 | tail -15 > gpurun_out/s2_pytest.txt; tail -6 gpurun_out/s2_pytest.txt
bash tools/kdur2.sh s2_base
ACE_CONV_WL=1 bash tools/kdur2.sh s2_wl
for t in base wl; do echo "== $t"; grep "conv_w\|steps/s" gpurun_out/kdur_s2_$t.txt | cut -c1-150; done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 > gpurun_out/s2_bench_plain.json 2>gpurun_out/s2_bench_plain.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --hooks > gpurun_out/s2_bench_hooks.json 2>gpurun_out/s2_bench_hooks.err
ACE_NO_FUSED_PHYSICS=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --hooks > gpurun_out/s2_bench_hooks_torch.json 2>gpurun_out/s2_bench_hooks_torch.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f16x3 --hooks --graph window > gpurun_out/s2_bench_hooks_window.json 2>gpurun_out/s2_bench_hooks_window.err
for f in plain hooks hooks_torch hooks_window; do python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/s2_bench_$f.json')); print('$f', d['value'], d['ms_per_step'])
except Exception as e: print('$f failed', e); print(open('gpurun_out/s2_bench_$f.err').read()[-800:])
"; done
