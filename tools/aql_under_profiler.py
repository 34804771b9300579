"""usage: python tools/aql_under_profiler.py [calls=400] [steps=20] — the own-queue decode loop of the tiny reference fixture, call after call, with progress
on stderr: under `rocprofv3 --kernel-trace` the queue is the profiler's intercept queue; says after how many packets anything goes wrong."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import booster_amd as bamd
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = bamd.Model(os.path.join(G, "tiny_a.gguf")); ctx = bamd.Context(m, 128)
first = None
for i in range(calls):
    ctx.decode([1, 2, 3, 4, 5, 6, 7, 8], 0)
    out, ms = ctx.generate_greedy(8, steps)
    if first is None: first = out.copy()
    assert np.array_equal(out, first)
    if i % 20 == 0: print("call", i, "aql_runs", ctx.aql_runs(), file=sys.stderr, flush=True)
print("done", calls, "calls,", ctx.aql_runs(), "on the own queue")
