#!/bin/bash
# Build a variant of libunetk.so with extra -D flags on one source: tools/build_ab.sh NAME SOURCE.hip -DFLAG...
# -> build_ab/libunetk_NAME.so (select it with UNETK_AB_LIB in the tools/bench_*.py scripts).  build_ab/ does not travel
# to the GPU box; UNETK_AB_DIR=<another directory> builds there for a run on the box (delete it afterwards).
set -e
cd "$(dirname "$0")/../unet_amd/csrc"
name=$1; src=$2; shift 2
D=${UNETK_AB_DIR:-build_ab}; mkdir -p ../../$D
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c "$src" -o ../../$D/${name}_${src%.hip}.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] || objs="$objs $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../$D/libunetk_${name}.so $objs ../../$D/${name}_${src%.hip}.o
echo built $D/libunetk_${name}.so
