#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
if [ "$PARITY" != "0" ]; then timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sweep.py -k "mul_mat_batch" -x -q 2>&1 | tail -4; fi
python tools/prefill_quick.py 512 2>&1 | tail -1
BAMD_PREFILL_WAVES=8 python tools/prefill_quick.py 512 2>&1 | tail -1
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 2>&1 | tail -1
for k in "$@"; do BAMD_LIB=booster_amd/lib/libbooster_amd_$k.so python tools/prefill_quick.py 512 2>&1 | tail -1; done
