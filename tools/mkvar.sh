#!/bin/bash
# builds exp/libexp_NAME.so with extra -D flags (compile-time variants for tools/ab.sh / ACE_SFNO_LIB)
# usage: mkvar.sh NAME -DFLAG=.. ...   -> exp/libexp_NAME.so
name=$1; shift
cd "$(dirname "$0")/.."; mkdir -p exp
python - "$name" "$@" <<'PY'
import sys
from ace_amd import build
name, extra = sys.argv[1], sys.argv[2:]
print("built", build.build(force=True, out=f"exp/libexp_{name}.so", extra=extra, objdir=f"exp/obj_{name}"))
PY
