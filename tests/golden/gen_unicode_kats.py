#!/usr/bin/env python3
"""tests/golden/unicode_classes.json: the code-point classes the reference's pre-tokeniser regexes see — \\p{L}, \\p{N}, \\s as
unicode_cpt_flags reports them (cpp/src/unicode.h:59, table cpp/src/unicode-data.cpp) — as [lo, hi] ranges, recorded by running the
genuine reference (oracle/_ref/unicode_ref = oracle/harness/unicode_ref.cpp linked against oracle/_ref/libggml_ref.so).  Data only.
tools/gen_unicode_tables.py generates booster_amd/csrc/bamd_unicode_tables.h from this file.  Build container only."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
exe = os.path.join(ROOT, "oracle", "_ref", "unicode_ref")
out = {}
for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines():
    p = line.split()
    out[p[0]] = [[int(x, 16) for x in r.split("-")] for r in p[1:]]
json.dump(out, open(os.path.join(HERE, "unicode_classes.json"), "w"), separators=(",", ":"))
print({k: (len(v), sum(b - a + 1 for a, b in v)) for k, v in out.items()})
