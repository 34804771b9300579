#!/bin/bash
# usage (GPU box): tools/aql_dbg.sh — the own-queue decode loop under rocprofv3 --kernel-trace across several wraps of the queue's ring (the tool's intercept queue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/aqldbg; rm -rf $O; mkdir -p $O
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/aql_under_profiler.py 400 20 ) > $O/kt.txt 2>&1; echo "kernel-trace rc $?"; grep -E "^call|done|SIGSEGV|bamd_aql" $O/kt.txt | tail -4
find $O -name '*.db' -delete; find $O -name '*_kernel_trace.csv' -delete
head -8 $(find $O -name '*kernel_stats.csv') | cut -c1-200
