"""Builds booster_amd/lib/libbooster_amd.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

    python -m booster_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libbooster_amd.so")
SOURCES = ["bamd_matvec.hip", "bamd_matvec_fast_a.hip", "bamd_matvec_fast_b.hip", "bamd_attention.hip", "bamd_attention_mfma.hip", "bamd_colaunch.hip", "bamd_prefill.hip", "bamd_prefill2.hip", "bamd_sampler.hip", "bamd_engine.cpp", "bamd_gguf.cpp", "bamd_vocab.cpp", "bamd_bridge.cpp"]
HEADERS = ["bamd_formats.h", "bamd_kernels.h", "bamd_device.h", "bamd_matvec_core.h", "bamd_attn_fused.h", "bamd_mfma_common.h", "bamd_gguf.h", "bamd_vocab.h", "bamd_unicode_tables.h", "../../include/bamd.h", "../../include/booster_bridge.h"]
# -ffp-contract=off: the numerics contract (bit-parity with the reference CPU path) forbids implicit FMA fusion.
# -fno-slp-vectorize (decode kernels only, NO_SLP): SLP packs neighbouring f32 multiplies into v_pk_mul_f32, whose operands need
# even-aligned register pairs: the copies it adds sit right behind the loads (a full s_waitcnt before the weight ring could be
# requested) and packed f32 is no faster there.  The prefill kernels keep SLP: their f32 chains run on float4 accumulators, where
# v_pk_fma_f32 halves the instruction count (same IEEE fma per element).
NO_SLP = ("bamd_matvec.hip", "bamd_matvec_fast_a.hip", "bamd_matvec_fast_b.hip", "bamd_attention.hip", "bamd_colaunch.hip")
# gfx950 kernarg preload for the decode mat-vec kernels: their leading scalar parameters (BAMD_LEAD_PARAMS: activation / weight pointers, K, eps)
# arrive in SGPRs at wave launch, so the first requests do not wait for an s_load of the argument block (+0.3 % decode, measured)
KERNARG_PRELOAD = {"bamd_matvec_fast_a.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"], "bamd_matvec_fast_b.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-unused-value"]


LIB_TIMING = os.path.join(HERE, "lib", "libbooster_amd_timing.so")      # -DBAMD_TIMING: kernels write phase stamps (tools/timeline.py)


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, timing=False, variant=None, extra=()):
    """variant / extra: experiment builds — lib/libbooster_amd_<variant>.so compiled with the extra flags (select with BAMD_LIB)"""
    lib = LIB_TIMING if timing else LIB
    if variant:
        lib = os.path.join(HERE, "lib", "libbooster_amd_%s.so" % variant)
    if not force and not _stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "lib", variant) if variant else os.path.join(HERE, "lib", "timing") if timing else os.path.join(HERE, "lib")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + (["-fno-slp-vectorize"] if s in NO_SLP else []) + KERNARG_PRELOAD.get(s, []) + (["-DBAMD_TIMING"] if timing else []) + list(extra) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    _v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, timing="--timing" in sys.argv, variant=_v, extra=[a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-f") or a.startswith("-m") or a.startswith("-amdgpu")]))
