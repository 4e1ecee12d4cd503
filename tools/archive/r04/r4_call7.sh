#!/bin/bash
# round 4, seventh GPU call: dhconv degrees dealt per XCD (A/B against the previous library), then the verification of the binary
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/kdur2.sh c7_prev $GRAFT_REPO_ROOT/exp/libexp_prev.so; grep "steps/s" gpurun_out/kdur_c7_prev.txt
bash tools/kdur2.sh c7_new; grep "steps/s" gpurun_out/kdur_c7_new.txt
grep -h "dhconv_strip" gpurun_out/kdur_c7_prev.txt gpurun_out/kdur_c7_new.txt
bash tools/r4_verify.sh r04v2 pmc noquarter
