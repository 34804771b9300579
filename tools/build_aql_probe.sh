#!/bin/bash
# builds tools/aql_probe (+ its raw gfx950 code object for the HSA path); cross-compiles without a GPU
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/aql_probe.hip -o tools/aql_probe -lhsa-runtime64
/opt/rocm/bin/hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 -std=c++17 tools/aql_probe.hip -o tools/aql_probe.hsaco
