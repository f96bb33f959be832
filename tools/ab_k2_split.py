'''The convolutions with an appended shortcut phase (K2) of one full-size CFG forward: the rule against ping-pong tiles x split-K (the rule keeps
them on the 2-barrier kernels; the sweeps never tried split-K with K2).  usage: ab_k2_split.py'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_recorder
from flexdiffuse_amd import hip
rec, keep = gemm_recorder.record('sd15', 64, 8, vae=False)
st, lib = hip.stream(), hip.lib()


def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for key, (d, cnt) in rec.items():
    k = dict(zip(gemm_recorder.KEY_FIELDS, key))
    if not (k['conv'] and k['K2']):
        continue

    def run(tile, sk):
        d.tile, d.split_k = tile, sk
        return lib.fd_gemm_f16(ctypes.byref(d), st)
    run(0, 0)
    cands = [(0, 0)] + [(t, s) for t in (30, 32, 33, 13, 16) for s in (1, 2, 4, 8) if run(t, s) == 0]
    torch.cuda.synchronize()
    best = {}
    for _ in range(3):
        for c in cands:
            best[c] = min(best.get(c, 1e9), timeit(lambda: run(*c)))
    d.tile, d.split_k = 0, 0
    top = sorted(best.items(), key=lambda kv: kv[1])[:6]
    print(f"M={k['M']:6d} N={k['N']:5d} K={k['K']:5d} {gemm_recorder.describe(key):22s} x{cnt:2d}: rule {best[(0, 0)]:6.1f}  " +
          ' '.join(f'{c}:{v:.1f}' for c, v in top), flush=True)
