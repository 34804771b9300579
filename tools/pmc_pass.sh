#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" <cmd...>   — one rocprofv3 --pmc pass (own run, no trace flags), CSV reduced to per-kernel means
tag=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
(cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $ctr --output-format csv -d $out -- "$@") > $out/log.txt 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv; log tail:"); print(open(sys.argv[1] + "/log.txt").read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if any(t in k for t in ("matvec", "attn", "mfma", "quantize", "matmul")):
        print(k[:70], {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
