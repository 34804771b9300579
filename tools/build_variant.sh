#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [flags...]  — lib/libbooster_amd_<name>.so = the main build with ONE source recompiled with extra flags
set -e
R=$(cd "$(dirname "$0")/.." && pwd); L=$R/booster_amd/lib; n=$1; f=$2; shift 2
mkdir -p $L/$n
o=$L/$n/$(basename ${f%.*}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-result -Wno-unused-value "$@" -c $R/booster_amd/csrc/$f -o $o
objs=""; for x in $L/*.o; do [ "$(basename $x)" = "$(basename $o)" ] && objs="$objs $o" || objs="$objs $x"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libbooster_amd_$n.so $objs
echo $L/libbooster_amd_$n.so
