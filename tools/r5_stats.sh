#!/bin/bash
# usage: tools/r5_stats.sh <variant lib name or "main"> [n=512]: rocprofv3 kernel stats of the prefill for one library
R=${GRAFT_REPO_ROOT:-$(pwd)}; v=$1; n=${2:-512}
[ "$v" != "main" ] && export BAMD_LIB=booster_amd/lib/libbooster_amd_$v.so
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r5st_$v; rm -rf $O; mkdir -p $O
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tools/prefill_profile.py $n ) > $O.out 2>&1 < /dev/null
f=$(find $O -name '*kernel_stats.csv' | head -1); echo "== $v"; [ -n "$f" ] && head -8 "$f" | cut -c1-150
find $O -name '*.db' -delete; find $O -name '*_kernel_trace.csv' -delete
