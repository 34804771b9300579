cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --steps 128 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config']['repeats']['tokens_per_s'])"; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=1
run AMD_OPT_FLUSH=0
run GPU_FLUSH_ON_EXECUTION=0
run ROC_USE_FGS_KERNARG=0
run AMD_DIRECT_DISPATCH=0
run X=2
