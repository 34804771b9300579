"""CPU, world_size 2 / 4 / 8 (gloo): the layer-split schedule of booster_amd.pipeline — message order, token feedback, several
sequences in flight — with a CPU stand-in for the GPU stage.  The stand-in is deterministic integer arithmetic, so the
pipelined result must equal a sequential single-process evaluation."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from booster_amd.pipeline import FakeStage      # the deterministic CPU stand-in lives in the package (bench.py --backend gloo uses it too)


def reference(prompt, n_decode, n_layers):
    st = FakeStage(list(range(n_layers)), True, True)
    out, tok = [], None
    for pos in range(len(prompt) + n_decode):
        t = prompt[pos] if pos < len(prompt) else tok
        if pos >= len(prompt):
            out.append(t)
        st.step(0, t, None, pos, None, None, True, 0)
        tok = st.last_tok[0]
    return out


def worker(rank, world, port, n_seq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from booster_amd import pipeline
    ranges = pipeline.split_layers(5, world)
    st = FakeStage(list(range(*ranges[rank])), rank == 0, rank == world - 1)
    fed = pipeline.run_pipeline(st, dist, rank, world, [3, 1, 4, 1, 5], 7, n_seq)
    # the bench's two-phase use: 4 steps, then continue from the last fed token at its own position (pos_offset)
    fed1 = pipeline.run_pipeline(st, dist, rank, world, [3, 1, 4, 1, 5], 4, n_seq)
    carry = [fed1[0][-1] if rank == 0 else 0]
    fed2 = pipeline.run_pipeline(st, dist, rank, world, carry, 3, n_seq, pos_offset=5 + 3)
    if rank == 0:
        q.put((fed, [a + b for a, b in zip(fed1, fed2)]))
    dist.barrier()
    dist.destroy_process_group()


LONG_PROMPT = [(7 * i + 3) % 97 for i in range(1100)]


def worker_long(rank, world, port, n_seq, q):
    """a 1100-token prompt: three micro-batches (512 + 512 + 76) per sequence through the batched prompt phase, then 5 decode steps"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from booster_amd import pipeline
    ranges = pipeline.split_layers(5, world)
    st = FakeStage(list(range(*ranges[rank])), rank == 0, rank == world - 1)
    fed = pipeline.run_pipeline(st, dist, rank, world, LONG_PROMPT, 5, n_seq)
    calls = list(st.prefill_calls)
    if rank == 0:
        q.put((fed, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_seq", [1, 2, 3])
def test_layer_split_world2(n_seq):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_seq + (os.getpid() % 200)
    procs = [ctx.Process(target=worker, args=(r, 2, port, n_seq, q)) for r in range(2)]
    for p in procs: p.start()
    fed, fed_two_phase = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = reference([3, 1, 4, 1, 5], 7, 5)
    assert len(fed) == n_seq and len(fed_two_phase) == n_seq
    for f in fed + fed_two_phase:
        assert f == want


@pytest.mark.parametrize("n_seq", [1, 2])
def test_long_prompt_in_microbatches_world2(n_seq):
    """the prompt crosses the RCCL pipeline in micro-batches of <= 512 positions (one [T, n_embd] message per boundary each),
    not one token per round; the generated tokens equal the sequential evaluation"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + n_seq + (os.getpid() % 200)
    procs = [ctx.Process(target=worker_long, args=(r, 2, port, n_seq, q)) for r in range(2)]
    for p in procs: p.start()
    fed, calls = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    want = reference(LONG_PROMPT, 5, 5)
    assert len(fed) == n_seq
    for f in fed:
        assert f == want
    assert calls == [(s, T, i0) for s in range(n_seq) for i0, T in ((0, 512), (512, 512), (1024, 76))]


def worker_deep(rank, world, port, n_seq, n_layers, prompt, n_decode, q):
    """BASELINE config 4's schedule: `n_layers` layers over `world` stages split by split_layers_balanced (the stage that also runs the
    output layer gets fewer), n_seq sequences in flight"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from booster_amd import pipeline
    ranges = pipeline.split_layers_balanced(n_layers, world, head_cost=1.3)
    st = FakeStage(list(range(*ranges[rank])), rank == 0, rank == world - 1)
    fed = pipeline.run_pipeline(st, dist, rank, world, prompt, n_decode, n_seq)
    calls = list(st.prefill_calls)
    if rank == 0:
        q.put((fed, calls, ranges))
    dist.barrier()
    dist.destroy_process_group()


def run_deep(world, n_seq, n_layers, prompt, n_decode, port_base):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + 10 * world + n_seq + (os.getpid() % 200)
    procs = [ctx.Process(target=worker_deep, args=(r, world, port, n_seq, n_layers, prompt, n_decode, q)) for r in range(world)]
    for p in procs: p.start()
    fed, calls, ranges = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return fed, calls, ranges


@pytest.mark.parametrize("world,n_seq", [(4, 1), (4, 4), (8, 1), (8, 3), (8, 8)])
def test_eighty_layers_over_four_and_eight_stages(world, n_seq):
    """the schedule BASELINE config 4 runs — 80 layers over 8 stages (7 boundaries; also 4), 1..8 sequences in flight (fewer than,
    equal to and — at world 4 — as many as the stages) — gives the tokens of the sequential evaluation.  No schedule beyond world 2 had run anywhere before."""
    prompt = [3, 1, 4, 1, 5, 9, 2, 6]
    fed, calls, ranges = run_deep(world, n_seq, 80, prompt, 6, 29700)
    assert ranges[0][0] == 0 and ranges[-1][1] == 80 and len(ranges) == world
    if world == 8:
        assert ranges[-1] == (71, 80) and ranges[0] == (0, 10)           # 10 layers per stage, 9 + the output layer on the last (head cost 1.3 layers)
    want = reference(prompt, 6, 80)
    assert len(fed) == n_seq
    for f in fed:
        assert f == want
    assert calls == [(s, 8, 0) for s in range(n_seq)]                   # the 8-token prompt crossed every boundary as ONE [8, n_embd] block per sequence


def test_long_prompt_in_microbatches_world8():
    """a 1100-token prompt = three micro-batches (512 + 512 + 76) per sequence pipelined through EIGHT stages (micro-batch j + 1 enters stage r while j is in r + 1),
    two sequences, then decode rounds: tokens == sequential"""
    fed, calls, _ = run_deep(8, 2, 80, LONG_PROMPT, 4, 29800)
    want = reference(LONG_PROMPT, 4, 80)
    assert len(fed) == 2
    for f in fed:
        assert f == want
    assert calls == [(s, T, i0) for s in range(2) for i0, T in ((0, 512), (512, 512), (1024, 76))]


def test_prompt_microbatches():
    from booster_amd import pipeline
    assert pipeline.prompt_microbatches(5) == [(0, 5)]
    assert pipeline.prompt_microbatches(512) == [(0, 512)]
    assert pipeline.prompt_microbatches(1100) == [(0, 512), (512, 512), (1024, 76)]


def test_split_layers():
    from booster_amd import pipeline
    assert pipeline.split_layers(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert pipeline.split_layers(32, 1) == [(0, 32)]
    r = pipeline.split_layers(80, 8)
    assert r[0][0] == 0 and r[-1][1] == 80 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    for L, n in ((32, 1), (32, 2), (32, 4), (32, 8), (80, 8), (8, 8), (9, 8)):
        b = pipeline.split_layers_balanced(L, n)
        assert len(b) == n and b[0][0] == 0 and b[-1][1] == L and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(y > x for x, y in b)
    assert pipeline.split_layers_balanced(32, 8)[-1] == (29, 32)           # the stage that also runs lm_head gets fewer layers
    # the head's cost comes from the model's own bytes: ~1.5 layers of decode time at the 8B widths, ~1.3 at the 70B widths
    assert 1.4 < pipeline.head_cost_layers(144.5e6, 430.9e6) < 1.65 and 1.15 < pipeline.head_cost_layers(524.6e6, 861.9e6) < 1.4
    assert pipeline.split_layers_balanced(80, 8, head_cost=pipeline.head_cost_layers(524.6e6, 861.9e6))[-1] == (71, 80)
