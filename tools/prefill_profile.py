"""Only the batched prefill (MFMA path), a few repetitions — target for rocprofv3 --kernel-trace --stats (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import booster_amd as b
from booster_amd import gguf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
path = "/dev/shm/bamd_prefill_8b.gguf"
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=4096, H=32, Hkv=8, L=32, F=14336, V=128256, seed=7, reuse_layers=True)
m = b.Model(path); ctx = b.Context(m, 2048 if n <= 2044 else 4096)
toks = [(7919 * i + 13) % 128256 for i in range(n)]
for _ in range(4):
    for i in range(0, n, 512):
        ctx.decode(toks[i:i + 512], i)
