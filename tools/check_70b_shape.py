"""One-off parity check at Llama-3-70B layer width (E 8192, F 28672, H 64, Hkv 8, attn_v in Q5_K as in 70B Q4_K_M) with 2 layers and a
small vocabulary: batched prefill (MFMA + integer-dot kernels) and decode steps against the CPU oracle, bit for bit (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import booster_amd as b
from booster_amd import gguf
from oracle import pyoracle as po
path = "/dev/shm/bamd_70b_2l.gguf"
def tf(name, il):
    if name == "output": return gguf.Q6_K
    if name == "attn_v": return gguf.Q5_K if il else gguf.Q6_K
    if name == "ffn_down": return gguf.Q6_K if il == 0 else gguf.Q4_K
    return gguf.Q4_K
t0 = time.time()
if not os.path.exists(path):
    gguf.write_synthetic_llama(path, E=8192, H=64, Hkv=8, L=2, F=28672, V=1024, seed=17, type_fn=tf)
print("gguf %.1f s" % (time.time() - t0))
r = gguf.GGUFReader(path)
om = po.OracleModel(r); oc = po.OracleContext(om, 64, nthreads=32)
m = b.Model(path); ctx = b.Context(m, 64)
prompt = [(7919 * i + 13) % 1024 for i in range(12)]
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
t0 = time.time(); lo = oc.decode(prompt, 0); print("oracle prefill %.1f s" % (time.time() - t0))
lg = ctx.decode(prompt, 0)
print("prefill bit-identical:", bool(np.array_equal(bits(lg), bits(lo))), "max |d| = %g" % np.abs(lg - lo).max())
n = len(prompt); ok = True
for s in range(4):
    t = int(np.argmax(lo)); lo = oc.decode([t], n); lg = ctx.decode([t], n); n += 1
    ok = ok and bool(np.array_equal(bits(lg), bits(lo)))
print("4 decode steps bit-identical:", ok)
