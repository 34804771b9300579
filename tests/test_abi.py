"""CPU: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol the headers declare,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def L():
    from booster_amd import build
    return C.CDLL(build.build())


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:bamd_|init|initContext|doInference|stopInference|status|promptEval|getPromptTokenCount|timing|getSeed)\w*)\s*\(", txt)))


def test_bamd_symbols_exported(L):
    syms = [s for s in declared_symbols("bamd.h") if s.startswith("bamd_")]
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "missing export: " + s


def test_no_cpu_fallback(L):
    import booster_amd
    if booster_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.ones(256, np.float32)
    with pytest.raises(booster_amd.BamdError):
        booster_amd.op_quantize_q8_K(x)
    with pytest.raises(booster_amd.BamdError):
        booster_amd.Model(os.path.join(ROOT, "tests", "golden", "tiny_a.gguf"))


def test_bridge_symbols_exported(L):
    """the nine cgo symbols of cpp/bridge.h:132-165"""
    syms = declared_symbols("booster_bridge.h")
    assert sorted(syms) == sorted(["init", "initContext", "doInference", "stopInference", "status", "promptEval", "getPromptTokenCount", "timing", "getSeed"])
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
