#!/bin/bash
# usage: tools/r5_var.sh [variant]  — parity of the prefill mat-muls + timing of one library (default: the main one) by Q4_K / Q5_K layout and against the round-2 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
[ -n "$1" ] && export BAMD_LIB=booster_amd/lib/libbooster_amd_$1.so
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py -k "mul_mat_batch" -x -q 2>&1 | tail -5
for w in 64 32 16 8; do BAMD_PREFILL_WAVES=$w python tools/prefill_quick.py 512 6 2>&1 | tail -1 | sed "s/^/layout=$w /"; done
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 6 2>&1 | tail -1
python tools/prefill_quick.py 2048 3 2>&1 | tail -1
