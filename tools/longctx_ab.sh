#!/bin/bash
# tools/longctx_ab.sh <outfile under gpurun_out> — decode at ~8000 cached positions (8B shapes, n_ctx 8192): own AQL queue vs hipGraph replays, and the per-kernel times of the
# long-sequence attention launches (rocprofv3 --kernel-trace --stats on the hipGraph path)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-longctx_ab.txt}
: > $O
for mode in "BAMD_AQL=0" "BAMD_AQL=1" "BAMD_AQL=0" "BAMD_AQL=1"; do
  echo "== $mode" | tee -a $O
  env $mode BAMD_AQL_VERBOSE=1 timeout 600 python tools/longctx_bench.py 7936 8192 2>&1 | tail -2 | tee -a $O
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/longctx_stats; mkdir -p $R/gpurun_out/longctx_stats
( cd $R && BAMD_AQL=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/longctx_stats -- python tools/longctx_bench.py 7936 8192 ) > /dev/null 2>&1
python3 - $R/gpurun_out/longctx_stats >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:14]:
        print("%-90s calls %6s  avg %9.2f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $R/gpurun_out/longctx_stats -name '*_kernel_trace.csv' -delete; find $R/gpurun_out/longctx_stats -name '*.db' -delete
tail -16 $O
