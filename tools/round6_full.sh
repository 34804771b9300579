#!/bin/bash
# usage (GPU box, repo root): tools/round6_full.sh — the GPU test suite, then the round's profile set (tools/profile_round.sh), then a kernel-statistics pass with the
# decode steps as hipGraph replays (BAMD_AQL=0) beside the default own-queue pass, so the two can be compared per kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r6full
( cd $R && timeout 1500 python -m pytest tests -m gpu -x -q ) > $R/gpurun_out/r6full/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $R/gpurun_out/r6full/pytest.txt
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $R/gpurun_out/r6full/smoke.txt 2>&1; tail -1 $R/gpurun_out/r6full/smoke.txt
bash $R/tools/profile_round.sh prof_r06
O=$R/gpurun_out/prof_r06; mkdir -p $O/stats_graph
cd /tmp && export TMPDIR=/tmp
( cd $R && BAMD_AQL=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_graph -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-secondary ) > $O/bench_under_rocprof_graph.json 2> $O/stats_graph.err < /dev/null
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*.db' -delete
for f in $(find $O/stats $O/stats_graph -name '*kernel_stats.csv'); do echo "== $f"; head -12 $f | cut -c1-160; done
