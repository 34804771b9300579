"""GPU, two or more REAL devices (skipped on a one-GPU box; the driver's multi-GPU node runs them):

  * the bridge's layer split (`gpus:` / BOOSTER_GPUS, llama.cpp:5932-5969) over real devices — peer copies of the hidden state ordered by
    hipEvents between the stage streams (ggml-cuda.cu:2360-2411) — must reproduce the genuine reference's ABI transcript, exactly as the
    one-device and virtual-device runs of tests/test_gpu_bridge.py do;
  * booster_amd.pipeline with one process per GPU over RCCL ("nccl" backend): the greedy tokens of a layer-split model must equal the
    single-GPU result, for one sequence and for several sequences in flight.
The same schedule is covered on CPU by tests/test_pipeline_gloo.py (world_size 2, gloo) and the split rule by tests/test_abi.py."""
import os
import sys

import numpy as np
import pytest

from booster_amd import gguf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_devices(bamd, n):
    if bamd.device_count() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, bamd.device_count()))


def test_bridge_split_over_two_real_devices(bamd, tmp_path, monkeypatch):
    _need_devices(bamd, 2)
    import test_gpu_bridge as tb
    monkeypatch.delenv("BAMD_VIRTUAL_DEVICES", raising=False)
    monkeypatch.setenv("BOOSTER_GPUS", "1,1")
    lib = tb.bind(bamd)
    tb.replay_transcript(lib, bamd, tmp_path, "two")


def _rank(rank, world, port, path, n_layer, n_seq, prompt, n_decode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import booster_amd
    from booster_amd import pipeline
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    lo, hi = pipeline.split_layers(n_layer, world)[rank]
    st = pipeline.HipStage(booster_amd, torch, path, rank, (lo, hi), rank == 0, rank == world - 1, 256, n_seq)
    fed = pipeline.run_pipeline(st, dist, rank, world, prompt, n_decode, n_seq)
    if rank == 0:
        q.put(fed)
    dist.barrier()
    st.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_seq", [(1, 2), (2, 1), (2, 2)])
def test_pipeline_ranks_rccl(bamd, tmp_path, world, n_seq):
    """world 1 runs on any GPU box (the stage object, its hipGraph steps on a side stream, the token feedback and the RCCL process
    group of one rank); world 2 needs two devices."""
    _need_devices(bamd, world)
    import torch.multiprocessing as mp
    path = str(tmp_path / "split.gguf")
    L, V = 4, 512
    gguf.write_synthetic_llama(path, E=512, H=4, Hkv=1, L=L, F=768, V=V, seed=5)
    prompt = [(7919 * i + 13) % V for i in range(9)]
    n_decode = 12
    m = bamd.Model(path); ctx = bamd.Context(m, 256)
    ctx.decode(prompt, 0)
    want, _ = ctx.generate_greedy(len(prompt), n_decode - 1)
    want = [int(t) for t in want[:n_decode]]
    ctx.close(); m.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29700 + 2 * world + n_seq + (os.getpid() % 200)
    procs = [mpc.Process(target=_rank, args=(r, world, port, path, L, n_seq, prompt, n_decode, q)) for r in range(world)]
    for p in procs: p.start()
    fed = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(fed) == n_seq
    for f in fed:
        assert [int(t) for t in f] == want, "layer split over RCCL differs from the single-GPU greedy tokens"


@pytest.mark.parametrize("model", ["8b", "70b"])
def test_bench_pipeline_smoke_on_one_gpu(model):
    """`bench.py --gpus 1 --pipeline-smoke [--model 70b]`: the N > 1 leg of the bench — layer-split stage object, hipGraph stage steps,
    the grouped send / recv rounds on an RCCL process group — driven end to end by a single rank, so that the code the driver runs on its
    8-GPU node is exercised on every one-GPU box.  The JSON line must carry the BASELINE metric, a positive value and the stage roofline."""
    import json
    import subprocess
    if model == "70b":
        st = os.statvfs("/dev/shm") if os.path.isdir("/dev/shm") else None
        if st is None or (st.f_bavail * st.f_frsize < (46 << 30) and not os.path.exists("/dev/shm/bamd_llama3_70b_q4_k_m_synth.gguf.done")):
            pytest.skip("no room for the 42 GB synthetic Llama-3-70B GGUF")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pipeline-smoke", "--model", model, "--steps", "8", "--warmup", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["metric"].startswith("decode tokens/sec") and line["value"] > 0 and line["n_gpus"] == 1
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
