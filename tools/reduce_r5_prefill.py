"""Reduce the counter passes of tools/r5_prefill_pmc.sh (three --pmc passes per kernel generation, each its own rocprofv3 run) to profiles/r05_prefill_pmc_summary.json.

    python tools/reduce_r5_prefill.py gpurun_out r5f profiles/r05_prefill_pmc_summary.json
"""
import csv, glob, json, os, sys
from collections import defaultdict

src, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]
NAMES = {"v1": "round-2 kernels (bamd_prefill.hip, BAMD_PREFILL_V=1): 8 waves, every wave expands its own 16 rows",
         "w8": "round-5 eight-wave kernels (bamd_prefill2.hip, BAMD_PREFILL_WAVES=8): fragments built once per 64 x 64 workgroup, 16 x 32 per wave",
         "w16": "round-5 sixteen-wave kernels (bamd_prefill2.hip, the default): one 16 x 16 tile per wave, Q6_K on the eight-wave kernel"}
out = {"command": "tools/r5_prefill_pmc.sh: per generation three passes of `rocprofv3 --pmc <8 counters> --output-format csv -- python tools/prefill_profile.py 512` "
                  "(512-token micro-batches of the synthetic Llama-3-8B Q4_K_M GGUF; trace and counter passes are separate runs)",
       "note": "means per dispatch.  SQ_* instruction counters are per wave-instruction; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs: clocks of a dispatch = GRBM_GUI_ACTIVE / 8.  mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (clocks x 1024 SIMDs); "
               "valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs); instr_per_mfma = (VALU + SALU + LDS + VMEM_RD) / MFMA",
       "generations": {}}
for v in ("v1", "w8", "w16"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_%s_%s_*" % (tag, v), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "matmul_mfma" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    gen = {}
    for k, cs in sorted(agg.items()):
        e = {c: int(sum(x) / len(x)) for c, x in sorted(cs.items())}
        e["dispatches"] = len(next(iter(cs.values())))
        clk = e.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if clk and e.get("SQ_INSTS_MFMA"):
            e["clocks_per_dispatch"] = int(clk)
            e["mfma_utilisation"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (clk * 1024.0), 4)
            e["valu_per_mfma"] = round(e.get("SQ_INSTS_VALU", 0) / e["SQ_INSTS_MFMA"], 2)
            e["instr_per_mfma"] = round((e.get("SQ_INSTS_VALU", 0) + e.get("SQ_INSTS_SALU", 0) + e.get("SQ_INSTS_LDS", 0) + e.get("SQ_INSTS_VMEM_RD", 0)) / e["SQ_INSTS_MFMA"], 2)
            e["lds_busy"] = round(e.get("SQ_LDS_IDX_ACTIVE", 0) / (clk * 256.0), 4)
        gen[k] = e
    if gen:
        out["generations"][v] = {"what": NAMES[v], "per_kernel": gen}
json.dump(out, open(dst, "w"), indent=1)
for v, g in out["generations"].items():
    for k, e in g["per_kernel"].items():
        print(v, k, "mfma", e.get("mfma_utilisation"), "valu/mfma", e.get("valu_per_mfma"), "instr/mfma", e.get("instr_per_mfma"), "lds", e.get("lds_busy"), "clk", e.get("clocks_per_dispatch"))
