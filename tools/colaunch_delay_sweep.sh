cd $GRAFT_REPO_ROOT
for d in 264 266 268 270 272 274; do
  echo -n "delay $d: "; BAMD_COLAUNCH_DELAY=$d timeout 300 python bench.py --steps 128 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['config']['repeats']['tokens_per_s'])"
done
BAMD_AQL=1 timeout 600 python tools/longctx_bench.py 7936 8192 2>&1 | tail -1
