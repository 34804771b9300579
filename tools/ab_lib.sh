#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <variant | main> ... — mvbench 8b mode-2 lines and one 8B bench line per library variant (python -m booster_amd.build --variant <v> -D...)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  L=""; [ "$v" != "main" ] && L=booster_amd/lib/libbooster_amd_$v.so
  echo "== $v"
  BAMD_LIB=$L python tools/mvbench.py 8b 2>/dev/null | grep -E "mode 2|gate" | grep -v prologue
  BAMD_LIB=$L python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['repeats']['tokens_per_s'], {k:v['us_per_launch'] for k,v in d['roofline']['per_kind'].items()})"
done
