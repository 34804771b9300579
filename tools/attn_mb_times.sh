#!/bin/bash
# per-micro-batch mean duration of the prefill attention kernel of a 2048-token prompt (GPU box): rocprofv3 kernel trace of tools/prefill_profile.py 2048
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; rm -rf $R/gpurun_out/pf2; mkdir -p $R/gpurun_out/pf2
(cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf2 -- python tools/prefill_profile.py ${1:-2048}) > /dev/null 2>&1
f=$(ls -t $R/gpurun_out/pf2/*/*kernel_trace.csv | head -1)
python3 - "$f" <<PY
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
att=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows if "attn_batch" in r["Kernel_Name"] and "kv_store" not in r["Kernel_Name"])
d=[x[1]/1e3 for x in att]; per=[[] for _ in range(4)]
for i,x in enumerate(d): per[(i//32)%4].append(x)
print("attention per micro-batch (us):", " ".join("%.1f"%(sum(p)/max(len(p),1)) for p in per))
tot=collections.defaultdict(float)
for r in rows: tot[r["Kernel_Name"].split("(")[0][:44]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,v in sorted(tot.items(),key=lambda kv:-kv[1])[:7]: print("%8.2f ms per prompt  %s"%(v/4,k))
PY
rm -rf $R/gpurun_out/pf2
