'''Round 6: the short-K transformer GEMMs of the forward (K <= 1280: the projections around the attentions; profiles/r06_gemm_loss_table.txt puts
~0.9 ms of the family's 2.5 ms loss per forward there) re-issued through the rule and with EVERY tile id the library has (incl. the ping-pong
tiles with a statistics epilogue, which the rule and tests/test_gpu_gemm_rule.py never try), interleaved, best of three.
    python tools/ab_short_k.py'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_recorder
from flexdiffuse_amd import hip
TILES = (2, 3, 9, 10, 12, 13, 14, 15, 16, 20, 30, 31, 32, 33)
rec, keep = gemm_recorder.record('sd15', 64, 8, vae=False)
st, lib = hip.stream(), hip.lib()


def _time(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for key, (d, cnt) in rec.items():
    k = dict(zip(gemm_recorder.KEY_FIELDS, key))
    if k['conv'] or k['K'] + k['K2'] > 1280 or k['act'] == 4 or k['M'] * max(k['batch'], 1) < 1024:
        continue

    def run(tile):
        d.tile, d.split_k = tile, 0
        return lib.fd_gemm_f16(ctypes.byref(d), st)
    assert run(0) == 0
    tile0, split0 = ctypes.c_int32(0), ctypes.c_int32(0)
    d.tile, d.split_k = 0, 0
    lib.fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile0), ctypes.byref(split0))
    cands = [t for t in TILES if run(t) == 0]
    torch.cuda.synchronize()
    best, auto = {}, []
    for _ in range(3):
        auto.append(_time(lambda: run(0), 8))
        for c in cands:
            best[c] = min(best.get(c, 1e9), _time(lambda: run(c), 6))
    d.tile, d.split_k = 0, 0
    top = sorted(best.items(), key=lambda kv: kv[1])[:4]
    print(f"M={k['M'] * max(k['batch'], 1):6d} N={k['N']:5d} K={k['K']:5d} {gemm_recorder.describe(key):22s} x{cnt:2d}: rule ({tile0.value},{split0.value}) {min(auto) * 1e3:6.1f} us | " +
          ', '.join(f'{c} {t * 1e3:.1f}' for c, t in top), flush=True)
