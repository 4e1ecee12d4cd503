#!/bin/bash
# builds exp/libexp_NAME.so from a copy of the kernel sources at HEAD (or the working tree with WORKTREE=1) with the given patches applied
# usage: mkpatched.sh NAME [patch ...]     -> exp/libexp_NAME.so
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; mkdir -p exp
wt=exp/wt_$name; rm -rf $wt; mkdir -p $wt
if [ -n "$WORKTREE" ]; then mkdir -p $wt/ace_amd $wt/include; cp -r ace_amd/csrc $wt/ace_amd/; rm -rf $wt/ace_amd/csrc/build; cp ace_amd/build.py ace_amd/__init__.py $wt/ace_amd/ 2>/dev/null || true; cp include/*.h $wt/include/
else git archive HEAD ace_amd/csrc ace_amd/build.py include | tar -x -C $wt; fi
for p in "$@"; do (cd $wt && patch -p1 --no-backup-if-mismatch < "$root/$p"); done
python - "$root" "$wt" "$name" <<'PY'
import importlib.util, sys, os
root, wt, name = sys.argv[1:]
spec = importlib.util.spec_from_file_location("wtbuild", os.path.join(root, wt, "ace_amd", "build.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print("built", m.build(force=True, out=os.path.join(root, "exp", f"libexp_{name}.so"), objdir=os.path.join(root, "exp", f"obj_{name}")), m.source_sha256()[:12])
PY
rm -rf $wt exp/obj_$name
