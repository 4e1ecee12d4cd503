#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_collect.sh r02f > gpurun_out/pmc_collect_r02f.log 2>&1
tail -5 gpurun_out/pmc_collect_r02f.log | cut -c1-300
bash tools/kdur2.sh final
head -16 gpurun_out/kdur_final.txt
