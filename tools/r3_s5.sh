#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q -k "sht or dft or fft or forward or parity or headline or quarter or shapes" 2>&1 | tail -8 > gpurun_out/s5_pytest.txt
bash tools/kdur2.sh s5_base
bash tools/kdur2.sh s5_ir16 exp/libexp_ir16.so
bash tools/kdur2.sh s5_fr32 exp/libexp_fr32.so
cat gpurun_out/s5_pytest.txt
for t in base ir16 fr32; do echo == $t; grep -E "steps/s|^dft" gpurun_out/kdur_s5_$t.txt; done
