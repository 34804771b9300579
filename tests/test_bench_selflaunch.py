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


def test_bench_gpus8_plumbing_line():
    """`python bench.py --gpus 8` — eight self-launched ranks, 80 layers over eight stages (seven boundaries), the two-phase timed use —
    and the N > 1 line carries every field of the bench contract, the group size as the process group itself reports it, and the tokens of the sequential evaluation"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "5", "--warmup", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 8 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "strong" and out["vs_baseline"] is None
    cfg = out["config"]
    assert cfg["gloo_ranks"] == 8 and cfg["group_world"] == 8 and cfg["group_allreduce_of_ones"] == 8.0
    assert len(cfg["layer_ranges"]) == 8 and cfg["layer_ranges"][0] == [0, 10] and cfg["layer_ranges"][-1] == [71, 80] and cfg["n_layer"] == 80
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"]) and set(("value", "unit", "cores", "kind")) <= set(out["cpu_baseline"])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_pipeline_gloo import reference
    assert cfg["fed_tokens"] == reference([3, 1, 4, 1, 5], 2 + 1 + 5, 80)
