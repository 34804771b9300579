#!/usr/bin/env python3
"""Per-phase timeline of the decode launches at the Llama-3-70B widths (GPU box; needs the timing build: `python -m booster_amd.build --timing`).

    BAMD_LIB=booster_amd/lib/libbooster_amd_timing.so python tools/timeline70.py [pos]

A four-layer model of the 70B widths (E 8192, 64 / 8 heads, F 28672; ffn_down Q4_K in layers 0-1, Q6_K in layers 2-3), one decode step; every launch of layers
1 and 3 with its phase stamps (tools/timeline.py has the phase legend), min / median / max over the workgroups, in us after the launch's first entry."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from timeline import reduce_one


def main():
    import booster_amd
    from booster_amd import gguf
    pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    path = "/dev/shm/bamd_70b_tl4.gguf"
    if not os.path.exists(path):
        gguf.write_synthetic_llama(path, E=8192, H=64, Hkv=8, L=4, F=28672, V=32000, seed=7,
                                   type_fn=lambda name, il: gguf.Q6_K if (name == "output" or (name == "ffn_down" and il >= 2)) else gguf.Q4_K)
    m = booster_amd.Model(path, device=0)
    ctx = booster_amd.Context(m, 512)
    prompt = [(7919 * i + 13) % 32000 for i in range(pos)]
    for i in range(0, pos, 128):
        ctx.decode(prompt[i:i + 128], i)
    tl = ctx.timeline_step(pos, replays=4)
    rows = []
    prev_end = None
    for i in range(tl.shape[0]):
        r = reduce_one(tl[i], None, prev_end)
        rows.append(r)
        if r is not None:
            prev_end = r["end"]
    valid = [i for i, r in enumerate(rows) if r is not None]
    per = (len(valid) - 2) // 4
    print("launches with stamps: %d (%d per layer)" % (len(valid), per))
    for n, i in enumerate(valid):
        r = rows[i]
        ph = " ".join("p%d %.1f/%.1f/%.1f" % (p, r["p%d_min" % p], r["p%d_med" % p], r["p%d_max" % p]) for p in range(1, 8) if "p%d_med" % p in r)
        full = tl[i].astype(np.float64); full[full == 0] = np.nan
        wx = ""
        if not np.all(np.isnan(full[:, 16:24])):
            wx = " | exit of waves 0..7: " + " ".join("%.1f" % ((x - r["start"]) / 100.0) for x in np.nanmedian(full[:, 16:24], axis=0))
        print("launch %2d  wgs %3d gap %5.2f span %6.2f skew %4.2f | %s%s" % (n, r["n_wg"], r["gap_us"] or 0.0, r["span_us"], r["entry_skew_us"], ph, wx))
    ctx.close(); m.close()


if __name__ == "__main__":
    main()
