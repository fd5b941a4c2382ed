#!/bin/bash
# Build a variant of libunetk.so with extra -D flags on one source: tools/build_ab.sh NAME SOURCE.hip -DFLAG...
# -> build_ab/libunetk_NAME.so (select it with UNETK_AB_LIB in the tools/bench_*.py scripts)
set -e
cd "$(dirname "$0")/../unet_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../build_ab
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c "$src" -o ../../build_ab/${name}_${src%.hip}.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] || objs="$objs $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libunetk_${name}.so $objs ../../build_ab/${name}_${src%.hip}.o
echo built build_ab/libunetk_${name}.so
