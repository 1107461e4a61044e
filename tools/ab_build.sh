#!/bin/bash
# A/B builds of one kernel file: tools/ab_build.sh NAME FILE.hip [-DFLAG ...] -> fiksi_amd/csrc/build/ab/libfiksi_amd_NAME.so
# (the other objects are the library's own; run a tool against it with FIKSI_AMD_LIBRARY=<that path>)
set -e
cd "$(dirname "$0")/../fiksi_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p build/ab
stem=$(basename "$src" .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c -o build/ab/${stem}_${name}.o -x hip "$src"
objs=""
for o in build/*.o; do
  if [ "$(basename $o .o)" = "$stem" ]; then objs="$objs build/ab/${stem}_${name}.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libfiksi_amd_${name}.so $objs
echo build/ab/libfiksi_amd_${name}.so
