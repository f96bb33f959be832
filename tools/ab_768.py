'''Round 6, VERDICT r5 next 4: the 768x768 configurations (c4: SD1.5 batch 4, c5: SD2.1-size batch 8; 96x96 latents: rows = 9 x 2^k).  Every
unique GEMM / convolution launch of one CFG forward is re-issued through the rule and with forced (tile, split_k) candidates -- the 2-barrier
tiles incl. 288x160 (23) and the ping-pong tiles 30..33 -- interleaved, best of three rounds; prints, per launch that matters, the rule's time and
the three best candidates, and the totals.
    python tools/ab_768.py [sd15|sd21] [batch]'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_recorder
from flexdiffuse_amd import hip
preset = sys.argv[1] if len(sys.argv) > 1 else 'sd15'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
TILES = (9, 12, 13, 14, 15, 16, 20, 23, 30, 31, 32, 33)
SPLITS = (1, 2, 4, 8)
rec, keep = gemm_recorder.record(preset, 96, batch, vae=False)
st, lib = hip.stream(), hip.lib()


def _time(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = []
for key, (d, cnt) in rec.items():
    k = dict(zip(gemm_recorder.KEY_FIELDS, key))

    def run(tile, sk):
        d.tile, d.split_k = tile, sk
        return lib.fd_gemm_f16(ctypes.byref(d), st)
    assert run(0, 0) == 0
    t_auto = _time(lambda: run(0, 0), 4)
    if t_auto * cnt < 0.05:
        continue
    tile0, split0 = ctypes.c_int32(0), ctypes.c_int32(0)
    d.tile, d.split_k = 0, 0
    lib.fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile0), ctypes.byref(split0))
    cands = []
    for tile in TILES:
        if (tile in (15, 31) and k['N'] % 256) or (tile in (16, 30, 32) and k['N'] % 320) or (tile >= 30 and (k['lno'] or k['trans'])):
            continue
        if k['act'] == 4 and tile not in (14, 15, 31):
            continue
        for sk in SPLITS:
            if sk > 1 and (k['act'] == 4 or k['batch'] > 1 or k['lnf'] or k['lno'] or ((k['K'] + k['K2']) // 64) // sk < 8 or
                           sk * k['M'] * k['N'] * 4 > d.workspace_bytes):
                continue
            if run(tile, sk) == 0:
                cands.append((tile, sk))
    torch.cuda.synchronize()
    best, auto = {}, []
    for _ in range(3):
        auto.append(_time(lambda: run(0, 0), 4))
        for c in cands:
            best[c] = min(best.get(c, 1e9), _time(lambda: run(*c), 3))
    d.tile, d.split_k = 0, 0
    top = sorted(best.items(), key=lambda kv: kv[1])[:3]
    rows.append((min(auto) * cnt, min(auto), (tile0.value, split0.value), top, key, cnt))
rows.sort(reverse=True)
tot_auto = sum(r[0] for r in rows)
tot_best = sum(min(r[1], r[3][0][1] if r[3] else r[1]) * r[5] for r in rows)
print(f'# {preset} 768x768 (96x96 latents), batch {batch} (CFG batch {2 * batch}): rule {tot_auto:.2f} ms per forward over {len(rows)} launch shapes, '
      f'best forced candidate per launch {tot_best:.2f} ms ({100 * (tot_auto / tot_best - 1):.1f} % to gain)')
for ta, a, rule, top, key, cnt in rows:
    k = dict(zip(gemm_recorder.KEY_FIELDS, key))
    print(f"M={k['M']:6d} N={k['N']:5d} K={k['K']:5d}{('+%d' % k['K2']) if k['K2'] else '':6s} {gemm_recorder.describe(key):22s} x{cnt:2d}: rule {rule} {a * 1e3:7.1f} us | " +
          ', '.join(f'{c} {t * 1e3:.1f}' for c, t in top), flush=True)
