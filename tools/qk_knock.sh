#!/bin/bash
# tools/qk_knock.sh <outfile> — what the long-sequence score launch is made of: timing-only knock-out builds (python -m booster_amd.build --variant qkkN -DBAMD_QK_KNOCK=N;
# 1 = no K requests, 2 = no chains, 4 = no score stores; results wrong, clocks meaningful), rocprofv3 per-kernel averages at ~8000 cached positions, hipGraph path
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-qk_knock.txt}
: > $O
cd /tmp && export TMPDIR=/tmp
for v in ${QK_VARIANTS:-main qkk1 qkk2 qkk4 qkk7}; do
  if [ $v = main ]; then unset BAMD_LIB; else export BAMD_LIB=$R/booster_amd/lib/libbooster_amd_$v.so; [ -f $BAMD_LIB ] || continue; fi
  D=$R/gpurun_out/qk_knock_$v; rm -rf $D; mkdir -p $D
  ( cd $R && BAMD_AQL=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python tools/longctx_bench.py 7936 8192 ) > $D/out.txt 2>&1
  python3 - $D $v >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
line = [l for l in open(sys.argv[1] + "/out.txt") if l.startswith("decode at")]
print("== %s  %s" % (sys.argv[2], line[-1].strip() if line else "?"))
if f:
    for r in csv.DictReader(open(f[0])):
        if "attn_qk" in r["Name"] or "attn_spv" in r["Name"]:
            print("   %-60s calls %6s  avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $D
done
cat $O
