#!/bin/bash
# usage (GPU box): tools/ab70.sh "ENV=1 ENV2=x" ... — one whole-70B bench line (config 4 at N = 1) per argument, each with that environment; "" = defaults
cd $GRAFT_REPO_ROOT
one() { env $1 python bench.py --model 70b --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', d['value'], d['ms_per_step'], {k:v['us_per_launch'] for k,v in d['roofline']['per_kind'].items()})"; }
for e in "$@"; do one "$e"; done
