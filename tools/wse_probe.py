"""Weight-stream engine bring-up probe (GPU box): each case runs in its own process under a timeout, so a fault in one does not take the others.

    python tools/wse_probe.py            all cases
    python tools/wse_probe.py --case N   one case (what the driver process spawns)

Cases: LDS-DMA facts of the device; single-piece mat-vecs through the engine kernel against the launch kernels (bit for bit) with the launch's
wall-clock span from the timeline stamps; then the whole 8B fixture through the engine (tests/test_gpu_wse.py does the same under pytest)."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, type, rows, K, norm, residual, pair)
MV = [("gate/up q4k 14336x4096 norm pair", 12, 14336, 4096, 1, 0, 1), ("qkv q4k 6144x4096 norm", 12, 6144, 4096, 1, 0, 0),
      ("wo q4k 4096x4096 +res", 12, 4096, 4096, 0, 1, 0), ("down q6k 4096x14336 +res", 14, 4096, 14336, 0, 1, 0),
      ("down q4k 4096x14336 +res", 12, 4096, 14336, 0, 1, 0), ("v q5k 1024x4096 norm", 13, 1024, 4096, 1, 0, 0),
      ("lm_head q6k 32000x4096 norm", 14, 32000, 4096, 1, 0, 0), ("odd rows q4k 4100x4096", 12, 4100, 4096, 0, 0, 0)]
NCS = (10, 12)


def case_selftest():
    import booster_amd as b
    print("selftest", [hex(int(v)) for v in b.wse_selftest()])


def case_mv(i, nc, thin=0):
    import numpy as np
    import booster_amd as b
    from booster_amd.gguf import random_kquant_tensor
    name, t, rows, K, norm, res, pair = MV[i]
    rng = np.random.default_rng(100 + i)
    W = random_kquant_tensor(t, K, rows, rng, amp=4.0)
    Wu = random_kquant_tensor(t, K, rows, rng, amp=4.0) if pair else None
    x = (rng.standard_normal(K) * 2).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32) if norm else None
    r = rng.standard_normal(rows).astype(np.float32) if res else None
    if pair:
        want = b.op_ffn_gate_up(t, W, Wu, rows, K, x, norm_w=nw, eps=1e-5)
    else:
        want = b.op_mul_mat_vec(t, W, rows, K, x, norm_w=nw, eps=1e-5, residual=r)
    t0 = time.time()
    try:
        got, info, tl = b.op_wse_matvec(t, W, rows, K, x, norm_w=nw, eps=1e-5, residual=r, w_up_raw=Wu, nc=nc, thin=thin, timeline=True)
    except b.BamdError as e:
        print("%-36s nc %2d: ERROR %s" % (name, nc, e)); return
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    st = tl[tl > 0]
    span = (int(tl.max()) - int(st.min())) / 100.0 if st.size else -1
    ev = []
    for k in range(8):
        col = tl[:, :, k]; col = col[col > 0]
        ev.append("%.1f" % ((int(col.max()) - int(st.min())) / 100.0) if col.size else "-")
    print("%-36s nc %2d thin %d: %s  mismatches %d/%d  span %.1f us  last-event-by-kind(us) %s  info %s  (%.1fs)" % (
        name, nc, thin, "OK " if bad.size == 0 else "BAD", bad.size, rows, span, " ".join(ev), list(info[:9]), time.time() - t0))
    if bad.size:
        print("    first bad rows", bad[:12], got[bad[:4]], want[bad[:4]])


def main():
    if "--case" in sys.argv:
        k = int(sys.argv[sys.argv.index("--case") + 1])
        if k == 0:
            case_selftest()
        else:
            k -= 1
            case_mv(k % len(MV), NCS[(k // len(MV)) % len(NCS)], thin=k // (len(MV) * len(NCS)))
        return
    n = 1 + len(MV) * len(NCS) + len(MV)        # last block: thin = 1 at nc = first
    for k in range(n):
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
            out = p.stdout.decode().strip().splitlines()
            keep = [l for l in out if "amdgpu.ids" not in l]
            print("\n".join(keep[-6:]) if keep else "(no output)", "" if p.returncode == 0 else "[rc %d]" % p.returncode, flush=True)
        except subprocess.TimeoutExpired:
            print("case %d: TIMEOUT" % k, flush=True)


if __name__ == "__main__":
    main()
