"""Which workgroups end a launch late?  Per launch kind (gate_up, down, qkv): the exit of every workgroup relative to the launch's first entry, averaged over the
layers, then grouped by blockIdx % 8 (the XCD a workgroup lands on under round-robin dispatch) and by blockIdx // 32 (position in the grid).
    BAMD_LIB=booster_amd/lib/libbooster_amd_timing.so python tools/wg_skew.py [pos]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import booster_amd


def main():
    pos = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    path = bench.model_path(); bench.ensure_model(path, 0)
    m = booster_amd.Model(path, device=0); ctx = booster_amd.Context(m, 512)
    prompt = [(7919 * i + 13) % bench.CFG_8B["V"] for i in range(pos)]
    for i in range(0, pos, 128):
        ctx.decode(prompt[i:i + 128], i)
    runs = []
    for rep in range(3):
        tl = ctx.timeline_step(pos, replays=3).astype(np.float64)       # [launches][512][24]
        runs.append(tl)
    L = m.n_layer
    per = (runs[0].shape[0] - 1) // L
    names = ["qkv", "attn+wo", "gate_up", "down"] if per == 4 else ["l%d" % i for i in range(per)]
    for j, nm in enumerate(names):
        ex, en = [], []
        for tl in runs:
            for il in range(L):
                a = tl[il * per + j].copy(); a[a == 0] = np.nan
                first = np.nanmin(a[:, [0, 8]])
                wg_exit = np.nanmax(a[:, 16:24], axis=1) - first           # [512]
                wg_entry = np.nanmin(a[:, [0, 8]], axis=1) - first
                ex.append(wg_exit); en.append(wg_entry)
        ex = np.array(ex) / 100.0; en = np.array(en) / 100.0
        n = int(np.sum(~np.isnan(ex[0])))
        mean_exit = np.nanmean(ex, axis=0)[:n]; mean_entry = np.nanmean(en, axis=0)[:n]
        print("== %s: %d workgroups; exit mean %.2f us, min %.2f, max %.2f (per-launch max %.2f on average); entry mean %.2f max %.2f" % (
            nm, n, mean_exit.mean(), mean_exit.min(), mean_exit.max(), np.nanmean(np.nanmax(ex, axis=1)), mean_entry.mean(), mean_entry.max()))
        idx = np.arange(n)
        print("   by blockIdx %% 8 : " + " ".join("%.2f" % mean_exit[idx % 8 == k].mean() for k in range(8)))
        print("   by blockIdx // 32: " + " ".join("%.2f" % mean_exit[idx // 32 == k].mean() for k in range((n + 31) // 32)))
        print("   entry by // 32  : " + " ".join("%.2f" % mean_entry[idx // 32 == k].mean() for k in range((n + 31) // 32)))
        late = np.argsort(-mean_exit)[:12]
        print("   latest workgroups: " + " ".join("%d(%.2f)" % (w, mean_exit[w]) for w in late))
        # is lateness persistent (same workgroups every launch) or random?
        z = ex[:, :n] - np.nanmean(ex[:, :n], axis=1, keepdims=True)
        print("   std of a workgroup's mean lateness %.3f us vs std within a launch %.3f us" % (np.nanstd(np.nanmean(z, axis=0)), np.nanmean(np.nanstd(z, axis=1))))


if __name__ == "__main__":
    main()
