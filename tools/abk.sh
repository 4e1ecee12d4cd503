#!/bin/bash
# same-box A/B of library variants (tools/mkvar.sh) on the WHOLE bench step under rocprofv3 --kernel-trace:
# per variant the medians of the kernels matching PATTERN and the step time; variants alternate REPS times.
# usage (on the GPU box): abk.sh PATTERN REPS variant...     ("base" = the in-tree library)
pat=$1; reps=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in $(seq $reps); do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$GRAFT_REPO_ROOT/exp/libexp_$v.so; fi
  bash tools/kdur2.sh ab_${v}_$rep $lib > /dev/null 2>&1
  echo "== $v (rep $rep): $(grep -h '^steps/s' gpurun_out/kdur_ab_${v}_$rep.txt)"
  grep -h -E "$pat" gpurun_out/kdur_ab_${v}_$rep.txt | cut -c1-40,67-140
done; done
