#!/bin/bash
# tools/ik_variants.sh <outfile under gpurun_out> — what each class of sc1 access of the attention kernels costs, timed under ordinary HIP launches (BAMD_AQL=0;
# the variant libraries are NOT coherent on the own queue): python -m booster_amd.build --variant ik_<name> -DBAMD_IK_<CLASS>=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-ik_variants.txt}
: > $O
for v in ${IK_VARIANTS:-main ik_plain ik_st0 ik_qkv0 ik_kvld0 ik_kvst0 main}; do
  if [ $v = main ]; then unset BAMD_LIB; else export BAMD_LIB=booster_amd/lib/libbooster_amd_$v.so; [ -f $BAMD_LIB ] || continue; fi
  BAMD_AQL=0 timeout 300 python bench.py --steps 128 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); pk=d['roofline']['per_kind']
print('%-9s %8.2f tok/s  %.4f ms/step  | us/launch: ' % ('$v', d['value'], d['ms_per_step']) + '  '.join('%s %.2f' % (k, v['us_per_launch']) for k, v in pk.items()))" | tee -a $O
done
