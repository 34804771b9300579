#!/usr/bin/env python3
"""Per-phase timeline of one decode step (GPU box; needs booster_amd/lib/libbooster_amd_timing.so = `python -m booster_amd.build --timing`).

    BAMD_LIB=booster_amd/lib/libbooster_amd_timing.so python tools/timeline.py [pos] [out.json]

Every launch of the captured step graph writes the device wall clock (100 MHz) at fixed phase points from waves 0 and 7 of each
workgroup (bamd_device.h TL_STAMP).  Phases — mat-vec mode A (one wave per row-group): 0 entry, 1 ring requests issued, 2 activation
prologue done, 3 first chunk consumed, 4 last chunk consumed, 7 exit.  Mode B (split-K): 0 entry, 1 ring issued, 2 prologue done,
3 terms of the first batch parked, 4 past the barrier, 5 first chain done, 7 exit.  Attention: 0 entry, 1 RoPE + KV store, 2 scores,
3 softmax, 7 exit.  Co-launched attention + wo (bamd_colaunch.hip): the attention role as above; the wo role 0 entry, 1 ring issued, 2 flags seen,
3 terms parked, 4 past the barrier, 5 chain done, 7 exit — both relative to the launch's first entry.  Reported per launch kind (median over the layers): first entry -> {median, last} workgroup at each phase, in µs,
the gap from the previous launch's last exit to this launch's first entry, and the launch's span.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reduce_one(full_u64, t0=None, prev_end=None):
    """one launch (or the workgroups of one role of it): u64 [workgroups][24] -> dict; t0: time origin (default: its own first entry)"""
    full = full_u64.astype(np.float64)
    full[full == 0] = np.nan
    a = full[:, :16].reshape(-1, 2, 8).copy()
    if not np.all(np.isnan(full[:, 16:24])):
        with np.errstate(all="ignore"):
            a[:, 0, 7] = np.nanmax(full[:, 16:24], axis=1)            # exit: the LAST wave of the workgroup
    if np.all(np.isnan(a[:, :, 0])):
        return None
    first = np.nanmin(a[:, :, 0])
    if t0 is None:
        t0 = first
    end = np.nanmax(a[:, :, 7])
    r = dict(start=first, end=end, span_us=(end - t0) / 100.0, entry_skew_us=(np.nanmax(a[:, :, 0]) - first) / 100.0,
             gap_us=None if prev_end is None else (first - prev_end) / 100.0, n_wg=int(np.sum(~np.isnan(a[:, 0, 0]))))
    for ph in range(1, 8):
        col = a[:, :, ph]
        if np.all(np.isnan(col)):
            continue
        r["p%d_med" % ph] = (np.nanmedian(col) - t0) / 100.0
        r["p%d_max" % ph] = (np.nanmax(col) - t0) / 100.0
        r["p%d_min" % ph] = (np.nanmin(col) - t0) / 100.0
    return r


def reduce_timeline(tl, L):
    """tl: u64 [launches][512][24] -> list of dicts per launch"""
    rows = []
    prev_end = None
    for i in range(tl.shape[0]):
        r = reduce_one(tl[i], None, prev_end)
        rows.append(r)
        if r is not None:
            prev_end = r["end"]
    return rows


def main():
    import booster_amd
    import bench
    pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    path = bench.model_path()
    bench.ensure_model(path, 0)
    m = booster_amd.Model(path, device=0)
    ctx = booster_amd.Context(m, 512)
    prompt = [(7919 * i + 13) % bench.CFG_8B["V"] for i in range(pos)]
    for i in range(0, pos, 128):
        ctx.decode(prompt[i:i + 128], i)
    tl = ctx.timeline_step(pos, replays=4)
    L = m.n_layer
    rows = reduce_timeline(tl, L)
    n_valid = sum(r is not None for r in rows)
    per = (n_valid - 1) // L                                  # launches per layer: 5, fewer when roles share a launch (bamd_colaunch.hip)
    co_attn = os.environ.get("BAMD_COLAUNCH", "1") != "0"
    names = ["qkv"] + (["attn+wo"] if co_attn else ["attention", "wo"]) + ["gate_up", "down"]
    if len(names) != per:
        names = ["qkv", "attention", "wo", "gate_up", "down"] if per == 5 else ["launch%d" % i for i in range(per)]
    H = 32
    kinds = {}
    for i, r in enumerate(rows):
        if r is None:
            continue
        if i == per * L:
            k = "lm_head"
        else:
            il, j = divmod(i, per)
            k = names[j]
            if names[j] in ("qkv", "down"):                   # layers with a Q6_K attn_v / ffn_down stream more bytes
                from booster_amd.gguf import q4_k_m_type, Q6_K
                k += "_q6k" if q4_k_m_type("ffn_down", il, L) == Q6_K else "_q4k"
            if names[j] == "attn+wo":                         # the two roles of the shared launch, both against the launch's first entry
                cut, ra_n, rw_n = H, "  role attention", "  role wo"
                ra, rw = reduce_one(tl[i][:cut], r["start"]), reduce_one(tl[i][cut:], r["start"])
                if ra is not None:
                    kinds.setdefault(ra_n, []).append(ra)
                if rw is not None:
                    kinds.setdefault(rw_n, []).append(rw)
        kinds.setdefault(k, []).append(r)
    out = {}
    keys = ["gap_us", "span_us", "entry_skew_us"] + ["p%d_%s" % (p, s) for p in range(1, 8) for s in ("min", "med", "max")]
    print("%-14s %5s %7s %7s %6s | phases: min/med/max us after the first workgroup's entry" % ("launch", "n", "gap", "span", "skew"))
    for k, rs in kinds.items():
        agg = {}
        for key in keys:
            v = [r[key] for r in rs if r.get(key) is not None]
            if v:
                agg[key] = round(float(np.median(v)), 2)
        out[k] = dict(n=len(rs), **agg)
        ph = " ".join("p%d %.1f/%.1f/%.1f" % (p, agg["p%d_min" % p], agg["p%d_med" % p], agg["p%d_max" % p]) for p in range(1, 8) if "p%d_med" % p in agg)
        print("%-14s %5d %7.2f %7.2f %6.2f | %s" % (k, len(rs), agg.get("gap_us", 0.0), agg["span_us"], agg["entry_skew_us"], ph))
    # per-wave exit per launch kind: median over workgroups and layers of (exit of wave w - first entry of the launch)
    wexit = {}
    for i in range(tl.shape[0]):
        if rows[i] is None:
            continue
        k = "lm_head" if i == per * L else names[i % per]
        full = tl[i].astype(np.float64); full[full == 0] = np.nan
        if np.all(np.isnan(full[:, 16:24])):
            continue
        wexit.setdefault(k, []).append(np.nanmedian(full[:, 16:24], axis=0) - rows[i]["start"])
    for k, v in wexit.items():
        print("%-10s median exit of waves 0..7 (us): " % k + " ".join("%.1f" % (x / 100.0) for x in np.nanmedian(np.array(v), axis=0)))
    tot = (rows[-1]["end"] - rows[0]["start"]) / 100.0
    print("step (first entry -> last exit): %.1f us" % tot)
    out["step_us"] = tot
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    ctx.close(); m.close()


if __name__ == "__main__":
    main()
