"""Per-wave phase clocks of the sixteen-wave prefill mat-mul (a -DX_TIMING build: BAMD_LIB=booster_amd/lib/libbooster_amd_tim.so): one op-level launch,
workgroup (0, middle row block); columns = cycles per super-block step spent up to each stamp.  usage: python tools/prefill_phase.py [K=4096] [rows=14336] [T=512]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import booster_amd as b
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14336
T = int(sys.argv[3]) if len(sys.argv) > 3 else 512
L = b.lib()
dbg = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
L.bamd_prefill_dbg.argtypes = [C.c_void_p]; L.bamd_prefill_dbg.restype = None
L.bamd_prefill_dbg(dbg.data_ptr())
rng = np.random.default_rng(1)
Wb = rng.integers(0, 256, size=(rows, K // 256, 144), dtype=np.uint8)
Wb[:, :, 0:4] = np.frombuffer(np.array([0.01, 0.005], np.float16).tobytes(), np.uint8)      # finite d / dmin in every block
X = rng.standard_normal((T, K)).astype(np.float32)
for _ in range(2):
    y = b.op_mul_mat_batch(12, Wb.reshape(-1), rows, K, X, impl=2)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(16, 8)
nb = max(int(d[0, 6]), 1)
names = ["d products", "e0-3+staging", "e4-7+build", "min terms", "vmcnt(0)", "barrier"]
print("shader clocks (s_memtime) per super-block step and wave, by phase (each stamp itself costs a scalar memory round trip: compare columns and waves, not absolutes); K=%d rows=%d T=%d, nb=%d" % (K, rows, T, nb))
print("wave " + " ".join("%12s" % n for n in names) + "        total")
for w in range(16):
    print("%4d " % w + " ".join("%12.1f" % (d[w, i] / nb) for i in range(6)) + " %12.1f" % (d[w, :6].sum() / nb))
print("mean " + " ".join("%12.1f" % (d[:, i].mean() / nb) for i in range(6)) + " %12.1f" % (d[:, :6].sum(axis=1).mean() / nb))
