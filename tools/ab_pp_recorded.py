'''Every unique fd_gemm_f16 launch of one full-size CFG forward (tools/gemm_recorder.py) through the rule and through the ping-pong tiles
30 / 31 / 32 / 33 (split 1): which launches the rule should hand to gemm_pp.hip.  usage: ab_pp_recorded.py [min us per forward = 20]'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_recorder
from flexdiffuse_amd import hip
rec, keep = gemm_recorder.record('sd15', 64, 8, vae=False)
st, lib = hip.stream(), hip.lib()
floor_us = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for key, (d, cnt) in rec.items():
    k = dict(zip(gemm_recorder.KEY_FIELDS, key))

    def run(tile, sk):
        d.tile, d.split_k = tile, sk
        return lib.fd_gemm_f16(ctypes.byref(d), st)
    run(0, 0)
    t0 = min(timeit(lambda: run(0, 0)) for _ in range(2))
    if t0 * cnt < floor_us:
        continue
    row = []
    for tile in (30, 31, 32, 33):
        if run(tile, 1) != 0:
            row.append('   -  ')
            continue
        t = min(timeit(lambda: run(tile, 1)) for _ in range(2))
        row.append(f'{t:6.1f}' + ('*' if t < 0.97 * t0 else ' '))
    d.tile, d.split_k = 0, 0
    print(f"M={k['M']:6d} N={k['N']:5d} K={k['K']:5d} {gemm_recorder.describe(key):26s} x{cnt:2d}: rule {t0:6.1f}   30/31/32/33: " + ' '.join(row), flush=True)
