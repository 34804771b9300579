"""Layer-split decode across GPUs, one process per GPU (Booster's `gpus:` split: llama.cpp:5932-5969; the hop the
reference does with cudaMemcpyPeerAsync + events: ggml-cuda.cu:2360-2411, ggml-backend.c:1751-1844).

Rank r owns a contiguous layer range (+ embedding on rank 0, output_norm + lm_head on the last rank) and the KV slice
of those layers.  Per token and boundary ONE point-to-point message moves the f32 hidden state [n_embd] to the next
rank (torch.distributed send/recv: RCCL over xGMI with backend "nccl", gloo in the CPU tests); the arg-max token
returns from the last rank to rank 0 the same way.  There is no collective: the path has no reduction.

With one sequence the stages run strictly one after another (the reference's batch-1 behaviour: tokens/s does not grow
with N).  With n_seq = N independent sequences in flight — Booster's pods — rank r works on sequence (tick - r) mod N,
so every GPU streams its weight slice all the time.

The schedule is host-side Python over a small `stage` object, so the same code runs on GPUs (HipStage) and in the
world_size-2 gloo tests (a CPU fake stage).
"""
import time


def split_layers(n_layer, n_stage):
    """Equal `gpus:` weights: stage i gets layers [round(L*i/n), round(L*(i+1)/n))."""
    cuts = [int(round(n_layer * (i + 1) / n_stage)) for i in range(n_stage)]
    return list(zip([0] + cuts[:-1], cuts))


# decode-time model of a launch sequence on the MI355X: weights stream at ~6.2 TB/s inside a launch, lm_head (one long launch) at
# ~6.4 TB/s, and every layer pays ~21 us that no byte explains (five dependent launches: boundary, first requests, chain tail; attention latency).
# Checked against the measured launch times: 8B 44.3 us per layer / 67 us lm_head, 70B 107 / 133.
STREAM_BPS, HEAD_BPS, LAYER_FIXED_S, HEAD_FIXED_S = 6.2e12, 6.4e12, 21e-6, 2e-6


def head_cost_layers(layer_bytes, head_bytes):
    """how many layers of decode time output_norm + lm_head cost on a model whose layers stream `layer_bytes` and whose output matrix is `head_bytes`"""
    return (head_bytes / HEAD_BPS + HEAD_FIXED_S) / (layer_bytes / STREAM_BPS + LAYER_FIXED_S)


def model_bytes(path):
    """(mean bytes of a layer's matrices, bytes of the output matrix) from the GGUF tensor table (the tied-embedding fallback included)"""
    from .gguf import GGUFReader
    r = GGUFReader(path)
    layer, n_layer, head = 0, 0, 0
    for name, t in r.tensors.items():
        nb = len(t["data"])
        if name.startswith("blk.") and name.endswith(".weight") and "norm" not in name:
            layer += nb
            n_layer = max(n_layer, int(name.split(".")[1]) + 1)
        elif name == "output.weight":
            head = nb
    if head == 0 and "token_embd.weight" in r.tensors:
        head = len(r.tensors["token_embd.weight"]["data"])
    return layer / max(n_layer, 1), head


def split_layers_balanced(n_layer, n_stage, head_cost=None, path=None):
    """`gpus:` weights chosen so that stage TIMES are even: the last stage also runs output_norm + lm_head, which costs `head_cost` layers of
    decode time — from the model's own bytes (path: its GGUF) through head_cost_layers: ~1.5 layers on the 8B shape (431 MB of Q6_K against
    ~146-161 MB per layer), ~1.3 on the 70B shape (862 MB against 524 MB per layer)."""
    if n_stage == 1:
        return [(0, n_layer)]
    if head_cost is None:
        head_cost = head_cost_layers(*model_bytes(path)) if path else 1.4
    target = (n_layer + head_cost) / n_stage
    cuts = [int(round(target * (i + 1))) for i in range(n_stage - 1)] + [n_layer]
    for i in range(n_stage - 1):                     # strictly increasing, at least one layer per stage (also the last one)
        lo = (cuts[i - 1] if i else 0) + 1
        cuts[i] = min(max(cuts[i], lo), n_layer - (n_stage - 1 - i))
    return list(zip([0] + cuts[:-1], cuts))


class HipStage:
    """One pipeline stage on one MI355X: booster_amd.Model slice + one Context (KV cache) per sequence in flight."""

    def __init__(self, booster_amd, torch, path, device, layer_range, is_first, is_last, n_ctx, n_seq):
        self.torch = torch
        self.is_first, self.is_last = is_first, is_last
        self.model = booster_amd.Model(path, device=device, layer_first=layer_range[0], layer_last=layer_range[1],
                                       with_embd=is_first, with_output=is_last)
        self.ctx = [booster_amd.Context(self.model, n_ctx) for _ in range(n_seq)]
        self.n_embd = self.model.n_embd
        self.device = torch.device("cuda", device)
        # a side stream: stage steps replay captured hipGraphs, and the legacy default stream cannot be captured
        self.stream = torch.cuda.Stream(device=self.device)
        self.time_steps = None                              # a list: step() brackets every stage step with a HIP event pair (bench roofline)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def new_hidden(self):
        return self.torch.zeros(self.n_embd, dtype=self.torch.float32, device=self.device)

    def new_token(self):
        return self.torch.zeros(1, dtype=self.torch.int32, device=self.device)

    def new_hidden_batch(self, n_tokens):
        return self.torch.zeros(n_tokens, self.n_embd, dtype=self.torch.float32, device=self.device)

    def prefill(self, seq, tokens, n_tokens, pos0, hin, hout, want_logits):
        """one prompt micro-batch (<= 512 tokens) through this stage: bamd_stage_prefill (the batched kernels: one pass over the stage's weights for
        the whole micro-batch); a shape without batched kernels is evaluated token by token with the same T > 1 semantics (same bits)"""
        stream = self.torch.cuda.current_stream().cuda_stream
        ok = self.ctx[seq].stage_prefill(tokens, n_tokens, pos0, None if hin is None else hin.data_ptr(), None if hout is None else hout.data_ptr(),
                                         want_logits, stream)
        if ok:
            return
        row = self.n_embd * 4
        for i in range(n_tokens):
            self.ctx[seq].stage_step(int(tokens[i]) if tokens is not None else 0, pos0 + i, None if hin is None else hin.data_ptr() + i * row,
                                     None if hout is None else hout.data_ptr() + i * row, want_logits and i == n_tokens - 1, 1, stream, None)

    def step(self, seq, token_host, token_dev, pos, hin, hout, want_logits, prefill_mode):
        stream = self.torch.cuda.current_stream().cuda_stream
        if self.time_steps is not None:
            a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            a.record()
        self.ctx[seq].stage_step(token_host, pos, None if hin is None else hin.data_ptr(), None if hout is None else hout.data_ptr(),
                                 want_logits, prefill_mode, stream, None if token_dev is None else token_dev.data_ptr())
        if self.time_steps is not None:
            b.record(); self.time_steps.append((a, b))

    def token_to(self, seq, token_dev):
        self.ctx[seq].stage_token_to(token_dev.data_ptr(), self.torch.cuda.current_stream().cuda_stream)

    def sync(self):
        self.torch.cuda.synchronize()

    def close(self):
        for c in self.ctx:
            c.close()
        self.model.close()


class FakeStage:
    """Deterministic CPU stand-in for a stage (the world_size-2 gloo tests and `bench.py --backend gloo`): hidden' = (hidden * 5 + pos + 11 * layer) mod 8191
    per layer; the first stage embeds the token, the last stage picks token = (sum(hidden) * 31 + 7) mod V as its 'arg-max'.  Integer arithmetic in
    f32, so a pipelined run must equal a sequential single-process evaluation exactly."""
    E, V = 16, 97

    def __init__(self, layers, is_first, is_last):
        import torch
        self.torch = torch
        self.layers, self.is_first, self.is_last = layers, is_first, is_last
        self.last_tok = {}
        self.prefill_calls = []                      # (seq, n_tokens, pos0) of every batched prompt micro-batch (the tests look at it)

    def new_hidden(self): return self.torch.zeros(self.E, dtype=self.torch.float32)
    def new_token(self): return self.torch.zeros(1, dtype=self.torch.int32)
    def new_hidden_batch(self, n_tokens): return self.torch.zeros(n_tokens, self.E, dtype=self.torch.float32)

    def _layers(self, h, pos):
        for l in self.layers:
            h = self.torch.remainder(h * 5 + pos + 11 * l, 8191)
        return h

    def step(self, seq, token_host, token_dev, pos, hin, hout, want_logits, prefill_mode):
        if self.is_first:
            tok = int(token_dev.item()) if token_dev is not None else token_host
            h = self.torch.arange(self.E, dtype=self.torch.float32) * 3 + tok
        else:
            h = hin.clone()
        h = self._layers(h, pos)
        if self.is_last:
            if want_logits:
                self.last_tok[seq] = int((int(h.sum().item()) * 31 + 7) % self.V)
        else:
            hout.copy_(h)

    def prefill(self, seq, tokens, n_tokens, pos0, hin, hout, want_logits):
        self.prefill_calls.append((seq, n_tokens, pos0))
        for i in range(n_tokens):
            self.step(seq, int(tokens[i]) if self.is_first else 0, None, pos0 + i, None if hin is None else hin[i], None if hout is None else hout[i],
                      want_logits and i == n_tokens - 1, 1)

    def token_to(self, seq, token_dev): token_dev.fill_(self.last_tok[seq])
    def sync(self): pass


# bench.py's "scaling" field is "strong" on every line (one sequence, the total work fixed as N grows); what the 1 -> 8 curve of a batch-1 layer split can look like:
SCALING_NOTE = ("batch-1 layer split: the N stages work one after another on ONE sequence, so value is flat (minus one hop per boundary) from 1 to 8 GPUs by "
                "construction - Booster's gpus: split buys capacity, not batch-1 speed; a flat curve is the expected result, not a failed strong-scaling run "
                "(pods_tokens_per_s is the rate with every stage busy)")

PREFILL_CAP = 512                                    # the reference's n_batch / n_ubatch: prompt positions per micro-batch (llama.cpp:16945-16960)


def prompt_microbatches(n_prompt, cap=PREFILL_CAP):
    return [(i, min(cap, n_prompt - i)) for i in range(0, n_prompt, cap)]


def run_pipeline(stage, dist, rank, world, prompt, n_decode, n_seq, pos_offset=0):
    """Greedy decode of n_seq sequences (same prompt) through `world` stages.  pos_offset: KV position of prompt[0] (a call can
    continue sequences whose first pos_offset positions are already in the stages' KV caches).

    Global schedule in ROUNDS: with period P = max(n_seq, world), rank r computes (sequence s, position pos) at round
    k = pos*P + s + r.  Every message is sent AND received in the same round — the hidden state of (s, pos) leaves rank r at
    the start of round k+1, where rank r+1 takes it; the arg-max of (s, pos) is held by the last rank until round
    (pos+1)*P + s, where rank 0 consumes it — and the operations of one round are issued as ONE batch_isend_irecv group.
    So rounds are identical exchange steps on all ranks and no cycle of blocked point-to-point operations can form
    (in-order RCCL streams or rendezvous gloo sends alike).  Positions < len(prompt) consume prompt tokens (known on rank 0),
    later positions the arg-max fed back from the last rank.

    A prompt of more than one token is evaluated in MICRO-BATCHES of <= 512 positions (`_prompt_phase`: the stage's batched kernels, ONE
    [T, n_embd] f32 message per boundary and micro-batch, micro-batch j + 1 entering stage r while j is in stage r + 1 — the reference's pipelined
    prompt evaluation, llama.cpp:16945-16960 / GGML_SCHED_MAX_COPIES ggml-backend.c:1030); the rounds above then start at the first generated
    position.  (A stage object without a `prefill` method keeps the one-token-per-round path for the prompt as well.)

    Returns, on rank 0, for every sequence the list of tokens FED after the prompt (n_decode of them); elsewhere empty lists.
    """
    import contextlib
    with (stage.stream_ctx() if hasattr(stage, "stream_ctx") else contextlib.nullcontext()):
        return _run_pipeline(stage, dist, rank, world, prompt, n_decode, n_seq, pos_offset)


def _run_pipeline(stage, dist, rank, world, prompt, n_decode, n_seq, pos_offset):
    n_prompt = len(prompt)
    total = n_prompt + n_decode                      # positions processed per sequence
    first, last = rank == 0, rank == world - 1
    P = max(n_seq, world)
    hin = [stage.new_hidden() for _ in range(n_seq)] if not first else [None] * n_seq
    hout = [stage.new_hidden() for _ in range(n_seq)] if not last else [None] * n_seq
    tok = [stage.new_token() for _ in range(n_seq)]          # rank 0: token being fed ; last rank: arg-max being held
    fed = [[] for _ in range(n_seq)]
    prefill_mode = 1 if n_prompt > 1 else 0
    last_round = (total - 1) * P + (n_seq - 1) + (world - 1) + 1
    start_pos = 0
    if n_prompt > 1 and hasattr(stage, "prefill"):
        _prompt_phase(stage, dist, rank, world, prompt, n_seq, pos_offset, tok, n_decode > 0)
        start_pos = n_prompt

    def slot(j):                                     # local index -> (pos, s) or None
        if j < 0:
            return None
        pos, s = divmod(j, P)
        return (pos, s) if s < n_seq and start_pos <= pos < total else None

    for k in range(start_pos * P, last_round + 1):
        ops = []
        if world > 1:
            # ---- sends of round k ----
            if not last:
                prev = slot(k - 1 - rank)            # what I computed in the previous round
                if prev is not None:
                    ops.append(dist.P2POp(dist.isend, hout[prev[1]], rank + 1))
            else:
                need = slot(k)                       # what rank 0 computes this round
                if need is not None and need[0] >= n_prompt:
                    ops.append(dist.P2POp(dist.isend, tok[need[1]], 0))
            # ---- receives of round k ----
            cur = slot(k - rank)
            if cur is not None:
                if not first:
                    ops.append(dist.P2POp(dist.irecv, hin[cur[1]], rank - 1))
                elif cur[0] >= n_prompt:
                    ops.append(dist.P2POp(dist.irecv, tok[cur[1]], world - 1))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        cur = slot(k - rank)
        if cur is None:
            continue
        pos, s = cur
        is_prompt = pos < n_prompt
        want_logits = last and (pos >= n_prompt - 1) and (pos + 1 < total)
        if first and not is_prompt:
            fed[s].append(tok[s].clone())
            tok_host, tok_dev = 0, tok[s]
        elif first:
            tok_host, tok_dev = int(prompt[pos]), None
        else:
            tok_host, tok_dev = 0, None
        stage.step(s, tok_host, tok_dev, pos + pos_offset, hin[s], hout[s], want_logits, prefill_mode if is_prompt else 0)
        if want_logits:
            stage.token_to(s, tok[s])                # last rank (world > 1: held until rank 0's round; world == 1: fed next)
    stage.sync()
    return [[int(t.item()) for t in f] for f in fed]


def _prompt_phase(stage, dist, rank, world, prompt, n_seq, pos_offset, tok, want_last):
    """The prompt of every sequence in micro-batches through the stages.  Item j = (sequence, first position, T); rank r evaluates item j in round
    j + r; the [T, n_embd] hidden block of item j leaves rank r at the start of round j + r + 1, where rank r + 1 takes it: sent and received in the
    same round, one batch_isend_irecv group per round and rank (the decode rounds' argument: no cycle of blocked point-to-point operations).  The
    last stage keeps the arg-max of each sequence's LAST prompt position in tok[s] (want_last), from where the first decode round sends it to rank 0."""
    first, last = rank == 0, rank == world - 1
    n_prompt = len(prompt)
    items = [(s, i0, T) for s in range(n_seq) for (i0, T) in prompt_microbatches(n_prompt)]
    t_max = max(T for _, _, T in items)
    hin = None if first else stage.new_hidden_batch(t_max)
    hout = None if last else stage.new_hidden_batch(t_max)
    for k in range(len(items) + world - 1):
        j = k - rank
        if world > 1:
            ops = []
            if not last and 0 <= j - 1 < len(items):
                ops.append(dist.P2POp(dist.isend, hout[:items[j - 1][2]], rank + 1))
            if not first and 0 <= j < len(items):
                ops.append(dist.P2POp(dist.irecv, hin[:items[j][2]], rank - 1))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        if 0 <= j < len(items):
            s, i0, T = items[j]
            want = last and want_last and i0 + T == n_prompt
            stage.prefill(s, prompt[i0:i0 + T] if first else None, T, i0 + pos_offset, None if hin is None else hin[:T], None if hout is None else hout[:T], want)
            if want:
                stage.token_to(s, tok[s])


def run_layer_split_bench(path, cfg, N, rank, local, prompt, n_ctx, warmup, steps, dist, torch, model_name="Llama-3-8B Q4_K_M"):
    """bench.py's N > 1 leg.  `value` = the BASELINE metric at N GPUs: greedy batch-1 decode of ONE sequence through the N layer-split
    stages (the stages work one after another, so it does not grow with N — Booster's `gpus:` split buys capacity, not batch-1 speed);
    the throughput with N independent sequences in flight (Booster's pods keeping every stage busy) is reported beside it."""
    import booster_amd
    ranges = split_layers_balanced(cfg["L"], N, path=path)
    stage = HipStage(booster_amd, torch, path, local, ranges[rank], rank == 0, rank == N - 1, n_ctx, N)
    dist.barrier()
    # untimed: every sequence through the prompt and `warmup` + 1 decode steps.  All sequences are identical, so the token fed at
    # the last position is known on rank 0; the timed calls re-feed it at the same position (T = 1 semantics, identical KV row)
    # and then run EXACTLY `steps` decode steps on top of the warm caches.
    fed = run_pipeline(stage, dist, rank, N, prompt, warmup + 1, N)
    pos0 = len(prompt) + warmup
    carry = [fed[0][-1] if rank == 0 else 0]
    # "RCCL saw N ranks", measured: the size of the process group and an all-reduce of ones over it
    ones = torch.ones(1, dtype=torch.float64, device=stage.device)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    rccl_world, rccl_sum = dist.get_world_size(), float(ones.item())
    # ---- prompt evaluation through the stages: the prompt of ONE sequence again (same positions, same KV rows), micro-batches of <= 512 positions, one
    #      [T, n_embd] send/recv per boundary and micro-batch (_prompt_phase) ----
    dist.barrier(); stage.sync()
    t0 = time.perf_counter()
    run_pipeline(stage, dist, rank, N, prompt, 0, 1)
    stage.sync(); dist.barrier()
    dt_prompt = time.perf_counter() - t0
    # ---- the metric: one sequence, K steps ----
    dist.barrier(); stage.sync()
    t0 = time.perf_counter()
    run_pipeline(stage, dist, rank, N, carry, steps, 1, pos_offset=pos0)
    stage.sync(); dist.barrier()
    dt_single = time.perf_counter() - t0
    # ---- side number: N sequences in flight ----
    dist.barrier(); stage.sync()
    t0 = time.perf_counter()
    run_pipeline(stage, dist, rank, N, carry, steps, N, pos_offset=pos0)
    stage.sync(); dist.barrier()
    dt_pods = time.perf_counter() - t0
    # ---- roofline per stage: this rank's slice of the weights / the time its stage step takes (events around 16 steps of a
    #      single sequence on the stage stream; an event pair's own cost, measured empty, is subtracted) ----
    stage.time_steps = []
    run_pipeline(stage, dist, rank, N, carry, 16, 1, pos_offset=pos0)
    stage.sync()
    ev = stage.time_steps; stage.time_steps = None
    with stage.stream_ctx():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); stage.sync()
        empty_ms = a.elapsed_time(b)
    stage_ms = max(sum(x.elapsed_time(y) for x, y in ev) / max(len(ev), 1) - empty_ms, 1e-6)
    wbytes = float(getattr(stage.model, "weight_bytes", 0))         # this rank's slice of the mat-mul weights
    t = torch.tensor([dt_single, dt_pods, dt_prompt], dtype=torch.float64, device=stage.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_single, dt_pods, dt_prompt = [float(v) for v in t.tolist()]
    per = torch.zeros(N, 2, dtype=torch.float64, device=stage.device)
    per[rank, 0] = wbytes; per[rank, 1] = stage_ms
    dist.all_reduce(per, op=dist.ReduceOp.SUM)
    per = per.tolist()
    stages = [dict(rank=r, layers=list(ranges[r]), weight_bytes=int(per[r][0]), ms_per_token=round(per[r][1], 4),
                   achieved_GBps=round(per[r][0] / (per[r][1] * 1e-3) / 1e9, 1)) for r in range(N)]
    slow = max(stages, key=lambda d: d["ms_per_token"])
    value = steps / dt_single
    stage.close()
    return dict(value=round(value, 2), ms_per_step=round(dt_single / steps * 1e3, 4), scaling="strong",
                cpu_baseline=dict(value=None, unit="tokens/s", cores=0, kind="reference", skipped="timed on rank 0 at N = 1 only (bench contract); see the N = 1 line"),
                config=dict(workload="%s shapes (synthetic GGUF), greedy batch-1 decode of ONE sequence, layer-split over %d MI355X "
                                     "(Booster's gpus: split), 128-token prompt, n_ctx %d" % (model_name, N, n_ctx),
                            parallelism="layer-split pp%d, one RCCL send/recv of the f32 hidden state [n_embd] per boundary per token" % N,
                            rccl_ranks=N, rccl_world=rccl_world, rccl_allreduce_of_ones=rccl_sum, layer_ranges=ranges,
                            prompt_eval_tokens_per_s=round(len(prompt) / dt_prompt, 1), prompt_microbatches=len(prompt_microbatches(len(prompt))),
                            sum_of_stage_ms=round(sum(d["ms_per_token"] for d in stages), 4),
                            pods_tokens_per_s=round(N * steps / dt_pods, 2),
                            note="value = one request through all stages (stages idle in turn: the reference's batch-1 behaviour); "
                                 "pods_tokens_per_s = N independent sequences in flight, every stage busy",
                            scaling_note=SCALING_NOTE),
                roofline=dict(bound="hbm", achieved=slow["achieved_GBps"], peak=8000.0, unit="GB/s", frac=round(slow["achieved_GBps"] / 8000.0, 4), traffic=None,
                              kernel="slowest stage: weight bytes of its layer slice / its stage time per token (HIP events on the stage stream)",
                              stages=stages))


def run_plumbing_check(N, rank, prompt, warmup, steps, dist, stage_cls, n_layer=None):
    """bench.py --backend gloo: everything of the N > 1 leg EXCEPT the GPU stage — launch, rendezvous, layer ranges, the round schedule with its
    two-phase use (warm-up, then K timed steps from the carried token), the group-size check, max-over-ranks timing, rank 0's JSON — on a deterministic
    CPU stand-in for the stage (FakeStage).  `value` is the stand-in's rate and means nothing; `config.fed_tokens` lets the test compare the pipelined
    tokens with a sequential evaluation.  N <= 2: five layers split evenly (the round-3 test's shape); beyond: 10 layers per rank split by
    split_layers_balanced, i.e. the 80-layer / 8-stage schedule of BASELINE config 4 at N = 8."""
    import torch
    if n_layer is None:
        n_layer = 5 if N <= 2 else 10 * N
    ranges = split_layers(n_layer, N) if N <= 2 else split_layers_balanced(n_layer, N, head_cost=1.3)
    stage = stage_cls(list(range(*ranges[rank])), rank == 0, rank == N - 1)
    dist.barrier()
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)                  # "the group saw N ranks", measured (the nccl leg reports the same two fields as rccl_*)
    fed = run_pipeline(stage, dist, rank, N, prompt, warmup + 1, 1)
    pos0 = len(prompt) + warmup
    carry = [fed[0][-1] if rank == 0 else 0]
    dist.barrier()
    t0 = time.perf_counter()
    fed2 = run_pipeline(stage, dist, rank, N, carry, steps, 1, pos_offset=pos0)
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    why = "plumbing check on CPU (gloo, stand-in stage): nothing is measured"
    return dict(value=round(steps / dt, 2), ms_per_step=round(dt / steps * 1e3, 4), scaling="strong",
                config=dict(workload="PLUMBING CHECK on CPU (gloo): the layer-split schedule over %d ranks with a deterministic stand-in stage — not a measurement" % N,
                            parallelism="layer-split pp%d" % N, gloo_ranks=N, group_world=dist.get_world_size(), group_allreduce_of_ones=float(ones.item()),
                            n_layer=n_layer, layer_ranges=ranges, fed_tokens=(fed[0] + fed2[0]) if rank == 0 else []),
                roofline=dict(bound="hbm", achieved=None, peak=8000.0, unit="GB/s", frac=None, traffic=None, skipped=why),
                cpu_baseline=dict(value=None, unit="tokens/s", cores=0, kind="reference", skipped=why))
