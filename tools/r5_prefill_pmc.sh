#!/bin/bash
# usage (GPU box): tools/r5_prefill_pmc.sh <tag>  — stall-reason counters of the prefill mat-mul kernels; VARIANTS = list of "name:env=val,env=val"
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r5pmc}
for v in ${VARIANTS:-v1:BAMD_PREFILL_V=1 w8:BAMD_PREFILL_WAVES=8 w16:BAMD_PREFILL_WAVES=16}; do
n=${v%%:*}; e=${v#*:}
for kv in ${e//,/ }; do export $kv; done
bash $R/tools/pmc_pass.sh ${T}_${n}_1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" timeout 300 python tools/prefill_profile.py 512 < /dev/null
bash $R/tools/pmc_pass.sh ${T}_${n}_2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_MFMA" timeout 300 python tools/prefill_profile.py 512 < /dev/null
bash $R/tools/pmc_pass.sh ${T}_${n}_3 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" timeout 300 python tools/prefill_profile.py 512 < /dev/null
for kv in ${e//,/ }; do unset ${kv%%=*}; done
done
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-v1:BAMD_PREFILL_V=1 w8:BAMD_PREFILL_WAVES=8 w16:BAMD_PREFILL_WAVES=16}; do
n=${v%%:*}; e=${v#*:}
for kv in ${e//,/ }; do export $kv; done
mkdir -p $R/gpurun_out/${T}_stats_$n
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_$n -- python tools/prefill_profile.py 512 ) > $R/gpurun_out/${T}_stats_$n.out 2>&1 < /dev/null
f=$(find $R/gpurun_out/${T}_stats_$n -name '*kernel_stats.csv' | head -1); echo "== $n"; [ -n "$f" ] && head -9 "$f" | cut -c1-160
for kv in ${e//,/ }; do unset ${kv%%=*}; done
done
find $R/gpurun_out -name '*.db' -delete; find $R/gpurun_out -name '*_kernel_trace.csv' -delete
find $R/gpurun_out -name '*counter_collection.csv' -size +20M -delete
