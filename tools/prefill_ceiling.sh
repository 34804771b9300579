#!/bin/bash
# tools/prefill_ceiling.sh <outfile under gpurun_out> — VERDICT r5 item 4: what the exact prefill mat-mul's eight per-lane sums per super-block cost.
# TIMING-ONLY builds of the sixteen-wave Q4_K kernel (python -m booster_amd.build --variant pceilN -DBAMD_PREFILL_CEILING=N; results are garbage by construction):
#   pceil1  the eight MFMAs of a super-block accumulate into ONE accumulator (K = 256), ONE chain FMA per super-block and tile; operand traffic, staging, fragment build unchanged
#   pceil2  the same + every B operand read feeds the MFMAs of two row tiles (1.5 KB of LDS reads per MFMA instead of 2; half the workgroups, the same MFMA count)
# one 512-token micro-batch of the 8B shape (tools/prefill_bench.py 512: 32 layers, lm_head for the last token), best of three
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-prefill_ceiling.txt}
: > $O
for v in main pceil1 pceil2 main; do
  if [ $v = main ]; then unset BAMD_LIB; else export BAMD_LIB=$R/booster_amd/lib/libbooster_amd_$v.so; [ -f $BAMD_LIB ] || continue; fi
  echo -n "$v: " | tee -a $O
  timeout 600 python tools/prefill_bench.py 512 2>/dev/null | grep "batched (MFMA)" | tee -a $O
done
cd /tmp && export TMPDIR=/tmp
for v in main pceil1 pceil2; do
  if [ $v = main ]; then unset BAMD_LIB; else export BAMD_LIB=$R/booster_amd/lib/libbooster_amd_$v.so; fi
  D=$R/gpurun_out/pceil_stats_$v; rm -rf $D; mkdir -p $D
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python tools/prefill_profile.py 512 ) > /dev/null 2>&1
  python3 - $D $v >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "matmul_mfma3" in r["Name"]:
            print("   %-8s %-60s calls %5s  avg %8.2f us" % (sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $D
done
cat $O
