#!/bin/bash
# same-box A/B of library variants built by tools/mkvar.sh (run on the GPU box: gpurun -- bash tools/ab.sh pmlp 10 -- base NAME)
# usage: ab.sh <microbench args...> -- variants...   (alternates variants twice within one call)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset ACE_LIB ACE_SFNO_LIB; else export ACE_SFNO_LIB=$GRAFT_REPO_ROOT/exp/libexp_$v.so ACE_LIB=$GRAFT_REPO_ROOT/exp/libexp_$v.so; fi
  rm -rf /tmp/ab_$v; rocprofv3 --kernel-trace --stats -f csv -d /tmp/ab_$v -o o -- python tools/microbench.py "${args[@]}" > /tmp/ab_$v.log 2>&1
  echo "== $v (rep $rep)"; grep -h "gemm[34]\|dft_" /tmp/ab_$v/o_kernel_stats.csv | awk -F'","' '{printf "%-90s n=%s avg=%.1f us\n", substr($1,2,90), $2, $4/1000}'
done; done
