#!/bin/bash
# tools/levers.sh <outdir under gpurun_out> — round 6, VERDICT r5 item 1: the decode levers as A/B lines of the same bench command (GPU box, from the repo root).
#   hipgraph      BAMD_AQL=0             one hipGraph per step on the context's HIP stream (rounds 1-5)
#   aql agent     BAMD_AQL_SCOPE=agent   the same packets from the library's own AQL queue, agent-scope fences between them (what HIP issues): the cost of the
#                                        graph replay itself (lever c)
#   aql none      (default)              fence scope NONE between the packets (lever a)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-levers}
mkdir -p "$O"
cd "$R"
run() { name=$1; shift; ( env "$@" BAMD_AQL_VERBOSE=1 timeout 300 python bench.py --steps 128 --warmup 16 --no-secondary --no-cpu-baseline > "$O/$name.json" 2> "$O/$name.err" ) < /dev/null
        python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["roofline"]["per_kind"]
    print("%-10s %8.2f tok/s  %.4f ms/step  repeats %s  aql_runs %s | us/launch: %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["repeats"]["tokens_per_s"], d["config"].get("aql_runs"),
          "  ".join("%s %.2f" % (n, v["us_per_launch"]) for n, v in k.items())))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for rep in 1 2; do
    run hipgraph_$rep BAMD_AQL=0
    run aql_agent_$rep BAMD_AQL_SCOPE=agent
    run aql_none_$rep BAMD_X=0
done
