#!/bin/bash
# usage (GPU box): tools/pmc_prefill.sh <tag>  — stall-reason counters of the batched-prefill kernels, three separate --pmc passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-x}
bash $R/tools/pmc_pass.sh ${T}_1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" timeout 300 python tools/prefill_profile.py 512 < /dev/null
bash $R/tools/pmc_pass.sh ${T}_2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" timeout 300 python tools/prefill_profile.py 512 < /dev/null
bash $R/tools/pmc_pass.sh ${T}_3 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" timeout 300 python tools/prefill_profile.py 512 < /dev/null
find $R/gpurun_out -name '*.db' -delete
