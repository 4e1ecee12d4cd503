#!/bin/bash
# variant library that differs from the in-tree build in ONE translation unit (seconds instead of minutes):
# usage: mkone.sh NAME file.hip -DFLAG=.. ...  -> exp/libexp_NAME.so  (the other objects come from ace_amd/csrc/build/, build the tree first)
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."; mkdir -p exp/obj_$name
python -c "from ace_amd import build; build.build()" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DACE_MEASUREMENT_SWITCHES "$@" -c ace_amd/csrc/$src -o exp/obj_$name/$src.o || exit 1
objs=$(ls ace_amd/csrc/build/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libexp_$name.so $objs exp/obj_$name/$src.o && echo built exp/libexp_$name.so
