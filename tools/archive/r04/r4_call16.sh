#!/bin/bash
# round 4, sixteenth GPU call: tile shape of the HEALPix packed convolutions (64 x 256 by the launcher's rule vs 128 x 128 forced)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in rule t128 rule2 t128b; do
  unset ACE_HPX_TILE_AB
  case $v in t128*) export ACE_HPX_TILE_AB=1;; esac
  timeout 300 python tools/bench_healpix.py --iters 30 > gpurun_out/r4_c16_healpix_$v.json 2> gpurun_out/r4_c16_healpix_$v.err; echo $v; cut -c95-250 gpurun_out/r4_c16_healpix_$v.json
done
export ACE_HPX_TILE_AB=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "healpix_unet" 2>&1 | tail -2
exit 0
