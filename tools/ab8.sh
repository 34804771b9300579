#!/bin/bash
# usage (GPU box): tools/ab8.sh "ENV=1" ... — one 8B bench line (the headline workload, no secondary legs, no CPU baseline) per argument, each with that environment
cd $GRAFT_REPO_ROOT
one() { env $1 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$1', d['value'], d['ms_per_step'], d['config']['repeats']['tokens_per_s'], {k:v['us_per_launch'] for k,v in d['roofline']['per_kind'].items()})"; }
for e in "$@"; do one "$e"; done
