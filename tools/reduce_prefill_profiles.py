"""Reduce the rocprofv3 output of the prefill profiling run (profiles/README.md) to the committed summaries.

    python tools/reduce_prefill_profiles.py gpurun_out/pprof profiles r01
"""
import csv, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges every call's output into the same directory: take the file of the LAST run"""
    return max(glob.glob(pattern), key=os.path.getmtime)

from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
shutil.copy(newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), os.path.join(dst, tag + "_prefill_kernel_stats.csv"))
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(newest(os.path.join(src, "pmc", "*", "*_counter_collection.csv")))):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE "
                  "--output-format csv -- python tools/prefill_profile.py 512",
       "note": "means per dispatch; GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); "
               "VALU utilisation = SQ_ACTIVE_INST_VALU x 4 cycles / the same denominator",
       "per_kernel": {}}
for k, cs in agg.items():
    e = {c: int(sum(v) / len(v)) for c, v in sorted(cs.items())}
    e["dispatches"] = len(next(iter(cs.values())))
    den = e.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0
    if den:
        e["mfma_utilisation"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / den, 4)
        e["valu_utilisation"] = round(e.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / den, 4)
    out["per_kernel"][k] = e
json.dump(out, open(os.path.join(dst, tag + "_prefill_pmc_summary.json"), "w"), indent=1)
for k, e in out["per_kernel"].items():
    if "mfma" in k or "attn" in k:
        print(k, e.get("mfma_utilisation"), e.get("valu_utilisation"))
