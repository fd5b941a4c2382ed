#!/bin/bash
# Timing-only variant of libunetk.so with extra -D flags on one source: tools/build_exp.sh NAME SOURCE.hip -DFLAG...
# -> exp_ab/libunetk_NAME.so (travels to the GPU box; select it with UNETK_AB_LIB in the tools/ scripts). Delete exp_ab/ afterwards.
set -e
cd "$(dirname "$0")/../unet_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../exp_ab
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c "$src" -o ../../exp_ab/${name}_${src%.hip}.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] || objs="$objs $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp_ab/libunetk_${name}.so $objs ../../exp_ab/${name}_${src%.hip}.o
rm ../../exp_ab/${name}_${src%.hip}.o
echo built exp_ab/libunetk_${name}.so
