#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for k in "$@"; do BAMD_LIB=booster_amd/lib/libbooster_amd_$k.so python tools/prefill_quick.py 512 6 2>&1 | tail -1; BAMD_PREFILL_WAVES=8 BAMD_LIB=booster_amd/lib/libbooster_amd_$k.so python tools/prefill_quick.py 512 6 2>&1 | tail -1;  done
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 6 2>&1 | tail -1
