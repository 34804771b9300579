"""CPU: `python bench.py --gpus 2` with NO launcher must start its own ranks (the driver starts the N > 1 line the way it starts the
N = 1 line).  Driven here on gloo with the deterministic stand-in stage of tests/test_pipeline_gloo.py (--backend gloo): launch, rendezvous, the layer-split
round schedule in its two-phase use, max-over-ranks timing and rank 0's single JSON line are the real code; only the GPU stage is replaced."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_launches_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["config"]["gloo_ranks"] == 2 and out["scaling"] == "strong"
    assert out["config"]["layer_ranges"] == [[0, 2], [2, 5]] or out["config"]["layer_ranges"] == [[0, 3], [3, 5]]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_pipeline_gloo import reference
    assert out["config"]["fed_tokens"] == reference([3, 1, 4, 1, 5], 2 + 1 + 6, 5)      # warm-up + 1 and the 6 timed steps, fed back through both ranks
