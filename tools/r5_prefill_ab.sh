#!/bin/bash
# usage (GPU box): tools/r5_prefill_ab.sh <outdir under gpurun_out>  — parity of the round-5 prefill kernel, then the A/B against the round-2 kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r5ab}
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py -k "mul_mat_batch" -x -q 2>&1 | tail -15 > "$O/parity.txt"
cat "$O/parity.txt"
for v in 1 2; do
  BAMD_PREFILL_V=$v timeout 300 python tools/prefill_bench.py 512 > "$O/pb512_v$v.txt" 2>&1; tail -5 "$O/pb512_v$v.txt"
  BAMD_PREFILL_V=$v timeout 300 python tools/prefill_bench.py 2048 > "$O/pb2048_v$v.txt" 2>&1; tail -5 "$O/pb2048_v$v.txt"
done
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  mkdir -p "$O/stats_v$v"
  ( cd "$R" && BAMD_PREFILL_V=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_v$v" -- python tools/prefill_profile.py 512 ) > "$O/stats_v$v.out" 2>&1 < /dev/null
  f=$(find "$O/stats_v$v" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
done
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*.db' -delete
