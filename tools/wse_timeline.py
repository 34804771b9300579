"""Per-edge timeline of the weight-stream engine on the 8B bench model (GPU box).

    python tools/wse_timeline.py [pos] [out.json]

One engine launch = the 32 layers of a decode step.  For every op of a layer (qkv, attention, wo, gate, up, down) the stamps of csrc/bamd_wse.h
(consumer 0: gather start, granules valid, activations ready, first / last record parked; chainer: first chunk chained, last row-group published;
loader: last slot issued), as offsets in us from the layer's start (median over CUs of the qkv gather start), median / min / max over the 256 CUs,
averaged over the middle layers.  This is the engine side of profiles/r04_engine_vs_launches.txt; the launch side is tools/timeline.py."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import booster_amd as b

EV = ["gather0", "valid", "actready", "firstrec", "lastrec", "chain0", "published", "loaded"]
OPS = ["qkv", "attn", "wo", "gate", "up", "down"]


def main():
    pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    b.set_wse(1)
    path = bench.ensure_model(bench.model_path("8b"), 0, "8b")
    m = b.Model(path); ctx = b.Context(m, bench.N_CTX)
    V = m.n_vocab
    ctx.decode([(7919 * i + 13) % V for i in range(128)], 0)
    for p in range(128, pos):
        ctx.decode([1], p)
    tl = ctx.wse_timeline(pos, replays=5).astype(np.int64)          # [n_cu][ops][8]
    n_cu, nops, _ = tl.shape
    L = nops // 6
    print("engine active:", ctx.wse_active(), " n_cu", n_cu, " ops", nops, " layers", L)
    t0 = tl[tl > 0].min()
    tot = (tl.max() - t0) / 100.0
    print("launch span (first stamp -> last stamp): %.1f us = %.2f us per layer" % (tot, tot / L))
    rows = {}
    lay = range(L // 4, 3 * L // 4)
    if int(os.environ.get("BAMD_WSE_THIN", "0")) & 2:          # experiment: slots 1 / 7 hold tick COUNTS (consumer 0 waiting for the loader / the chainer), not stamps
        for oi, on in enumerate(OPS):
            wf = np.mean([np.median(tl[:, li * 6 + oi, 1]) for li in lay]) / 100.0
            wr = np.mean([np.median(tl[:, li * 6 + oi, 7]) for li in lay]) / 100.0
            print("%-6s consumer 0 waited %.2f us for the loader, %.2f us for the chainer" % (on, wf, wr))
        tl[:, :, 1] = 0; tl[:, :, 7] = 0
    for li in lay:
        ref = np.median(tl[:, li * 6 + 0, 0][tl[:, li * 6 + 0, 0] > 0])
        for oi, on in enumerate(OPS):
            for ei, en in enumerate(EV):
                col = tl[:, li * 6 + oi, ei]; col = col[col > 0]
                if col.size == 0:
                    continue
                d = (col - ref) / 100.0
                rows.setdefault((on, en), []).append((np.median(d), d.min(), d.max(), col.size))
    nxt = []
    for li in lay:
        if li + 1 < L:
            a = np.median(tl[:, li * 6, 0][tl[:, li * 6, 0] > 0]); c = np.median(tl[:, (li + 1) * 6, 0][tl[:, (li + 1) * 6, 0] > 0])
            nxt.append((c - a) / 100.0)
    print("layer period (qkv gather start to the next layer's): %.2f us (min %.2f max %.2f)" % (np.mean(nxt), np.min(nxt), np.max(nxt)))
    out = {}
    print("%-6s %-10s %8s %8s %8s  %s" % ("op", "event", "median", "min", "max", "CUs"))
    for on in OPS:
        for en in EV:
            v = rows.get((on, en))
            if not v:
                continue
            a = np.array(v)
            print("%-6s %-10s %8.2f %8.2f %8.2f  %d" % (on, en, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), int(a[:, 3].mean())))
            out["%s.%s" % (on, en)] = [round(float(a[:, k].mean()), 3) for k in range(3)]
    if len(sys.argv) > 2:
        json.dump(dict(pos=pos, span_us=tot, layer_period_us=float(np.mean(nxt)), events=out), open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
