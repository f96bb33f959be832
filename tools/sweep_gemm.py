'''Record every fd_gemm_f16 launch of one CFG UNet forward (and one VAE decode) BY VALUE -- the descriptor with all of its
round-3 fields: appended operand (A2 / K2), LayerNorm fold, statistics emission, parity upsample (batch 4) -- keep every
buffer it points to alive, then re-issue each unique launch with every (tile, split_k) candidate and compare with the
library's own choice.
    python tools/sweep_gemm.py [preset = sd15] [latent = 64] [batch = 8] [--no-vae]
Outputs are overwritten in place (nothing reads them); launches the library refuses for a candidate are skipped.'''
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from flexdiffuse_amd import hip, ops, build
args = [a for a in sys.argv[1:] if not a.startswith('--')]
PRESET = args[0] if len(args) > 0 else 'sd15'
LAT = int(args[1]) if len(args) > 1 else 64
B = int(args[2]) if len(args) > 2 else 8
dev = torch.device('cuda:0')
import gemm_recorder
rec, keep = gemm_recorder.record(PRESET, LAT, B, vae='--no-vae' not in sys.argv, dev=dev)
orig_call = hip.call
print(f'{PRESET} latent {LAT} batch {B}: unique launches {len(rec)}', flush=True)


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


st = hip.stream()
tot_auto = tot_best = 0.0
for key, (d, cnt) in rec.items():
    (M, N, K, K2, conv, ih, iw, ic, kh, stride, up, act, trans, batch, of32, hasres, hasb2, lnf, lno, lda, ldc) = key

    def run(tile, sk):
        d.tile, d.split_k = tile, sk
        orig_call('fd_gemm_f16', ctypes.byref(d), st)
    t_auto = timeit(lambda: run(0, 0))
    row = {}
    for tile in (1, 2, 3, 4, 6, 9, 10, 11, 12, 13, 14, 15, 16, 20, 23, 30, 31, 32, 33):
        if act == 4 and tile in (2, 5, 7, 9, 12, 13, 16, 20, 23):
            continue
        if (tile == 15 and N % 256) or (tile == 16 and N % 320) or (tile == 23 and (M % 288 or N % 160)) or (tile >= 30 and (lno or trans)):
            continue
        for sk in (1, 2, 4, 8, 16):
            if sk > 1 and (act == 4 or batch > 1 or lnf or lno or K2 or ((K + K2) // 64) // sk < 4 or sk * M * N * 4 > d.workspace_bytes):
                continue
            try:
                row[(tile, sk)] = timeit(lambda: run(tile, sk), n=6)
            except Exception:
                continue
    if not row:
        continue
    best = min(row.items(), key=lambda kv: kv[1])
    fl = 2.0 * M * N * (K + K2) * max(batch, 1)
    tot_auto += t_auto * cnt; tot_best += min(best[1], t_auto) * cnt
    top = sorted(row.items(), key=lambda kv: kv[1])[:4]
    tag = ('conv ' if conv else 'gemm ') + ('up%d ' % up if up else '') + (f'+K2 {K2} ' if K2 else '') + ('LNfold ' if lnf else '') + \
          ('stats ' if lno else '') + ('V^T ' if trans else '') + ('res ' if hasres else '') + (f'x{batch} ' if batch > 1 else '') + ('GEGLU ' if act == 4 else '')
    flag = '  <<<' if best[1] < 0.93 * t_auto and (t_auto - best[1]) * cnt > 0.01 else ''
    print(f'M={M:6d} N={N:5d} K={K:5d} {tag:34s} x{cnt:3d}: auto {t_auto * 1e3:7.1f}us ({fl / t_auto / 1e9:5.0f}TF) best {best[0]} {best[1] * 1e3:7.1f}us  top: '
          + ' '.join(f'{k}:{v * 1e3:.0f}' for k, v in top) + flag, flush=True)
print(f'TOTAL per (unet fwd + vae decode): auto {tot_auto:.2f} ms, best-per-launch {tot_best:.2f} ms')
