cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -2
bash tools/ab8.sh "X=1" "X=2"
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc_null; rm -rf $O; mkdir -p $O
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary ) > $O/out.txt 2> $O/err.txt
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "matvec" in k or "attn_wo" in k: print("%-70s %6d  %.2f MB" % (k[:70], len(v), sum(v) / len(v) * 2048 / 1e6))
PY
