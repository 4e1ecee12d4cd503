#!/bin/bash
# same-box A/B of two whole trees (this one against a worktree of an earlier commit built in place: git worktree add exp/wt_rNN <commit>; build there):
# alternates `python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs` REPS times.  usage (GPU box): ab_rounds.sh exp/wt_r05 3
other=$1; reps=${2:-3}
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stages']
print('%-10s %7.2f steps/s  %6.3f ms  ' % (sys.argv[1], d['value'], d['ms_per_step']) + '  '.join('%s %.1f' % (k.split('.')[-1][:10], st[k]['us_per_launch']) for k in ('encoder','forward_transform.dft','forward_transform.legendre','dhconv','inverse_transform.legendre','inverse_transform.dft','inner_skip+activation','mlp.fc1','mlp.fc2+outer_skip','decoder')))" "$1"; }
for rep in $(seq $reps); do
  (cd $other && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null) | line "$(basename $other)"
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | line HEAD
done
