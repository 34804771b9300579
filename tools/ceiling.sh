#!/bin/bash
# usage (GPU box): tools/ceiling.sh <outfile under gpurun_out>  — decode rate of the shipped library against the TIMING-ONLY ceiling builds
# (python -m booster_amd.build --variant ceilN -DBAMD_CEILING=N; results of those builds are garbage by construction, only their clocks are read):
#   ceil1 = every mat-vec prologue without the Q8_K / RMSNorm statistics (plain load + LDS stores + one barrier)
#   ceil2 = split-K launches (QKV, wo, ffn_down) without the barrier + sequential chain replay
#   ceil3 = both
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r5_ceiling.txt}
: > $O
for v in main ceil1 ceil2 ceil3 main; do
  if [ $v = main ]; then unset BAMD_LIB; else export BAMD_LIB=booster_amd/lib/libbooster_amd_$v.so; [ -f $BAMD_LIB ] || continue; fi
  python bench.py --steps 128 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); pk=d['roofline']['per_kind']
print('%-6s %8.2f tok/s  %.4f ms/step  repeats %s  | us/launch: ' % ('$v', d['value'], d['ms_per_step'], d['config']['repeats']['tokens_per_s']) + '  '.join('%s %.2f' % (k, v['us_per_launch']) for k, v in pk.items()))" | tee -a $O
done
