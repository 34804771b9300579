#!/bin/bash
# usage (GPU box): tools/switch_matrix.sh — the GPU suite once per environment switch of INTEGRATION.md's table (every setting must give identical bits: the suite's
# reference pins run under each); one line per switch in gpurun_out/switch_matrix.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/switch_matrix.txt; : > $O
for e in "BAMD_AQL=0" "BAMD_AQL_SCOPE=agent" "BAMD_DOWN112=0 BAMD_GATEUP14=0 BAMD_QKV3=0 BAMD_WO4=0" "BAMD_PREFILL_AUX=0" "BAMD_MV_GENERIC=1" "BAMD_COLAUNCH=0 BAMD_GATEUP7=0 BAMD_DOWN14=0" "BAMD_ATTN_SPV=0 BAMD_ATTN_MFMA=0" "BAMD_JANUS_GPU=0 BAMD_STAGE_GRAPH=0" "BAMD_PREFILL_MFMA=0" "BAMD_MIXED_SPLIT=0 BAMD_COLAUNCH70=1 BAMD_QKV70_WAVES=8"; do
  r=$(env $e timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)
  echo "$e : $r" | tee -a $O
done
