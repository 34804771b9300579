"""Fixed cost of a layer-split stage step: the 8B synthetic model as 1, 2, 4, 8 virtual stages on ONE MI355X (hidden state handed over
as a device buffer exactly as between ranks, no RCCL), one sequence, greedy.  ms per token vs the number of stages (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import booster_amd as b
from booster_amd import gguf, pipeline
path = "/dev/shm/bamd_prefill_8b.gguf"
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, seed=7, reuse_layers=True)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    ranges = pipeline.split_layers(32, n)
    models = [b.Model(path, 0, r[0], r[1], i == 0, i == n - 1) for i, r in enumerate(ranges)]
    ctxs = [b.Context(m, 512) for m in models]
    hid = [torch.zeros(4096, dtype=torch.float32, device="cuda") for _ in range(n)]
    tdev = torch.zeros(1, dtype=torch.int32, device="cuda")
    def step(pos, tok_host, use_dev):
        for i in range(n):
            ctxs[i].stage_step(tok_host, pos, None if i == 0 else hid[i - 1].data_ptr(), None if i == n - 1 else hid[i].data_ptr(), i == n - 1, False, stream,
                               tdev.data_ptr() if (use_dev and i == 0) else None)
        ctxs[-1].stage_token_to(tdev.data_ptr(), stream)
    for pos in range(8):
        step(pos, 5, pos > 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 64
    for pos in range(8, 8 + K):
        step(pos, 0, True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%d stage(s): %.3f ms/token (host enqueue %.3f ms/token)" % (n, t_all / K * 1e3, t_host / K * 1e3))
    for c in ctxs: c.close()
    for m in models: m.close()
