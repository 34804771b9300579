cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py tests/test_gpu_fullsize_ref.py tests/test_f64_order.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
for v in spvx1 spvx2 spvx3 full; do
  echo "== $v"
  L=$GRAFT_REPO_ROOT/booster_amd/lib/libbooster_amd_$v.so; [ $v = full ] && L=$GRAFT_REPO_ROOT/booster_amd/lib/libbooster_amd.so
  BAMD_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lc_$v -o lc -- python tools/longctx_bench.py 7936 > gpurun_out/lc_$v.log 2>&1
  f=$(find gpurun_out/lc_$v -name "*kernel_stats.csv" | head -1)
  python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "attn_spv" in n or "attn_qk" in n: print(n[:40], r["Calls"], "avg ns", r["AverageNs"])
PY
done
for v in 1 0; do echo "== BAMD_ATTN_SPV=$v"; BAMD_ATTN_SPV=$v timeout 150 python tools/longctx_bench.py 7936 2>&1 | tail -1; done
