#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python tools/prefill_quick.py 512 6 2>&1 | tail -1
for k in "$@"; do BAMD_LIB=booster_amd/lib/libbooster_amd_$k.so python tools/prefill_quick.py 512 6 2>&1 | tail -1; done
BAMD_PREFILL_V=1 python tools/prefill_quick.py 512 6 2>&1 | tail -1
if [ -f booster_amd/lib/libbooster_amd_tim.so ]; then BAMD_LIB=booster_amd/lib/libbooster_amd_tim.so python tools/prefill_phase.py 4096 14336 512 2>/dev/null | tail -19; fi
