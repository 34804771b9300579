#!/bin/bash
# usage: tools/r5_var.sh <variant> — parity + timing of one variant lib, plus the phase clocks of its timing twin (tim)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BAMD_LIB=booster_amd/lib/libbooster_amd_$1.so
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py -k "mul_mat_batch" -x -q 2>&1 | tail -3
python tools/prefill_quick.py 512 2>&1 | tail -1
python tools/prefill_quick.py 2048 2>&1 | tail -1
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 2>&1 | tail -1
if [ -f booster_amd/lib/libbooster_amd_tim.so ]; then BAMD_LIB=booster_amd/lib/libbooster_amd_tim.so python tools/prefill_phase.py 4096 14336 512 2>/dev/null | tail -19; fi
