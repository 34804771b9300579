#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python tools/prefill_quick.py 512 6 2>&1 | tail -1
for k in "$@"; do BAMD_LIB=booster_amd/lib/libbooster_amd_$k.so python tools/prefill_quick.py 512 6 2>&1 | tail -1; done
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 6 2>&1 | tail -1
