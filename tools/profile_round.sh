#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_round.sh <outdir under gpurun_out>  — every rocprofv3 pass profiles/README.md lists, one call.
# Trace passes and --pmc passes are separate runs (never combined); everything is bounded by `timeout`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-prof}
rm -rf "$O"; mkdir -p "$O/stats" "$O/pmc" "$O/prefill/stats" "$O/prefill/pmc"
cd /tmp && export TMPDIR=/tmp
run() { ( cd "$R" && timeout 400 "$@" ) < /dev/null; }
run python bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"            # the full default run (config.secondary included): this is the line profiles/rNN_bench_n1.json holds
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-secondary > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc" -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > "$O/pmc.out" 2> "$O/pmc.err"
run rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prefill/stats" -- python tools/prefill_profile.py 512 > "$O/pstats.out" 2> "$O/pstats.err"
run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$O/prefill/pmc" -- python tools/prefill_profile.py 512 > "$O/ppmc.out" 2> "$O/ppmc.err"
run python tools/prefill_bench.py 512 > "$O/prefill_bench.txt" 2>&1
if [ -f "$R/booster_amd/lib/libbooster_amd_timing.so" ]; then
    ( cd "$R" && BAMD_LIB=booster_amd/lib/libbooster_amd_timing.so timeout 300 python tools/timeline.py 200 "$O/timeline.json" ) < /dev/null > "$O/timeline.txt" 2>&1
fi
# keep what travels back small: the per-dispatch traces are not needed, the stats and counter tables are
find "$O" -name '*_kernel_trace.csv' -delete
find "$O" -name '*.db' -delete
du -sh "$O"; cat "$O/bench_n1.json"; tail -2 "$O/prefill_bench.txt"; tail -3 "$O/timeline.txt" 2>/dev/null
