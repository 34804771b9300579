"""Reduce the rocprofv3 output of one profiling run (gpurun_out/prof, see profiles/README.md) to the committed summaries.

    python tools/reduce_profiles.py gpurun_out/prof profiles r01
"""
import csv, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges every call's output into the same directory: take the file of the LAST run"""
    return max(glob.glob(pattern), key=os.path.getmtime)

from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench_n1.json"), os.path.join(dst, tag + "_bench_n1.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
shutil.copy(newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), os.path.join(dst, tag + "_kernel_stats.csv"))

# PMC pass: FETCH_SIZE is reported in KB and, on gfx950, reads half of what a wide coalesced stream fetches (MI355X_MICROARCH.md, HBM)
agg = defaultdict(list)
for r in csv.DictReader(open(newest(os.path.join(src, "pmc", "*", "*_counter_collection.csv")))):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline",
       "correction": "bytes = FETCH_SIZE x 1024 (KB units) x 2 (gfx950 reports half of a wide coalesced stream; MI355X_MICROARCH.md, HBM section)",
       "per_kernel": {}}
mv_n, mv_b = 0, 0.0
for k, v in agg.items():
    mean = sum(v) / len(v)
    out["per_kernel"][k] = {"dispatches": len(v), "mean_FETCH_SIZE_raw": round(mean, 2), "mean_hbm_read_bytes": int(mean * 1024 * 2)}
    if k.startswith("matvec"):
        mv_n += len(v); mv_b += sum(v) * 1024 * 2
out["matvec_all"] = {"dispatches": mv_n, "mean_hbm_read_bytes_per_launch": int(mv_b / max(mv_n, 1))}
json.dump(out, open(os.path.join(dst, tag + "_pmc_fetch_summary.json"), "w"), indent=1)
print(json.dumps(out["matvec_all"]), {k: v["mean_hbm_read_bytes"] for k, v in out["per_kernel"].items() if "matvec" in k or "attn" in k})
