'''Round 6, VERDICT r5 next 6: an auditable loss table of the GEMM family over ONE CFG UNet forward of the bench workload (SD1.5, CFG batch 16,
64x64 latents; reference call site: the single unet(...) call of pipeline/guide.py:56-58).

Method.  (1) The forward's fd_gemm_f16 calls are recorded in launch order with their descriptors (the spy of tools/gemm_recorder.py) and the
(tile, split_k) the rule gives each (fd_gemm_plan).  (2) The same forward is replayed R times from its launch plan -- the launch mode the bench
measures -- with a HIP-event bracket around EVERY launch (hip.prof_*; the i-th GEMM bracket of a replay is the i-th recorded call); per position
the MEDIAN over the replays minus the empty-bracket cost is that launch's time in its real neighbourhood (caches as the forward leaves them; a
split-K launch's bracket holds its partial pass AND its finish pass).  (3) Per unique launch:
    FLOPs      2 M N (K + K2) x batch (the parity-decomposed upsample convolution at the MACs it issues)
    bytes      algorithmic: every operand once -- A (a convolution's input tensor once, not once per tap), W, the output, residual / appended rows
    floor      max(FLOPs / 1.2 PFLOP/s, bytes / 6.3 TB/s): the demonstrated MFMA rate of this part on activation-like data (DESIGN 3.9:
               power-limited; profiles/r05_pp_power.txt) and the demonstrated streaming rate (tools/micro/store_pattern.hip)
    loss       time - floor, x launches per forward
sorted by total loss.  The table's total time must reproduce the family's ms per forward (it is the same brackets summed).
    python tools/gemm_loss_table.py [replays = 12] > profiles/r06_gemm_loss_table.txt'''
import collections, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flexdiffuse_amd import build, hip, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
from tools.gemm_recorder import describe

R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
MFMA_FLOOR, HBM_FLOOR = 1.2e15, 6.3e12
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
t_dev = torch.full((1,), 400.0, device=dev)
unet.forward_nhwc(x, t_dev, ctx, rep=2)          # context K/V, one-time setup
torch.cuda.synchronize()

seq = []                                          # (key, desc copy, tile, split) in launch order
orig_call = hip.call


def spy(name, *a):
    if name == 'fd_gemm_f16':
        d = a[0]._obj
        key = (d.M, d.N, d.K, d.K2, d.conv, d.in_h, d.in_w, d.in_c, d.kh, d.stride, d.upsample2x, d.act, d.trans_out, d.batch,
               d.out_f32, bool(d.residual), bool(d.bias2), bool(d.ln_stats), bool(d.ln_stats_out), d.lda, d.ldc)
        c = ops.fd_gemm_desc()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(d))
        tile, split = ctypes.c_int32(0), ctypes.c_int32(0)
        c2 = ops.fd_gemm_desc()
        ctypes.memmove(ctypes.byref(c2), ctypes.byref(d), ctypes.sizeof(d))
        c2.gn_out = None                           # (the plan query is about the tile rule)
        hip.lib().fd_gemm_plan(ctypes.byref(c2), ctypes.byref(tile), ctypes.byref(split))
        seq.append((key + (bool(d.gn_out),), c, tile.value, split.value))
    return orig_call(name, *a)


pool = torch.cuda.MemPool()
plan = hip.Plan()
hip.call = spy
ops.hip.call = spy
try:
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        eps = unet.forward_nhwc(x, t_dev, ctx, rep=2)
finally:
    hip.call = orig_call
    ops.hip.call = orig_call
torch.cuda.synchronize()
for _ in range(3): plan.replay()
torch.cuda.synchronize()
# wall time of a forward without brackets
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): plan.replay()
e1.record(); torch.cuda.synchronize()
wall_ms = e0.elapsed_time(e1) / 10
empty_ms = hip.prof_calibrate(256)
hip.prof_set_stride(1)
hip.prof_enable(True)
for _ in range(R): plan.replay()
torch.cuda.synchronize()
hip.prof_enable(False)
fam, tag, ms, work, ex = hip.prof_drain(1 << 18)
n_all = len(fam) // R
assert n_all * R == len(fam), (len(fam), R)
fam, ms, work, ex = fam.reshape(R, n_all), ms.reshape(R, n_all), work.reshape(R, n_all), ex.reshape(R, n_all)
assert (fam == fam[0]).all()
med = np.maximum(np.median(ms, axis=0) - empty_ms, 0.0)        # per launch position, ms
fam0 = fam[0]
fam_ms = {name: float(med[fam0 == code].sum()) for code, name in ((0, 'gemm'), (1, 'attention'), (2, 'groupnorm'), (3, 'other'))}
gpos = np.nonzero(fam0 == 0)[0]
assert len(gpos) == len(seq), (len(gpos), len(seq))


def alg_bytes(d):
    batch = max(d.batch, 1)
    phase = d.conv and d.upsample2x == 2
    if d.conv:
        samples = d.M // (d.out_h * d.out_w)
        a = samples * d.in_h * d.in_w * d.in_c * 2
    else:
        a = d.M * d.K * 2 * batch
    a += d.M * d.K2 * 2
    w = d.N * (d.K + d.K2) * 2 * (batch if (d.batch_stride_w or phase) else 1)
    n_out = d.N // 2 if d.act == 4 else d.N
    rows = d.M * (4 if phase else batch)
    c = 0 if (d.gn_out and d.gn_skip_c) else rows * n_out * (4 if d.out_f32 else 2)
    c += rows * n_out * 2 if d.gn_out else 0
    r = rows * n_out * 2 if d.residual else 0
    st = d.M * 8 * ((1 if d.ln_stats else 0) + (1 if d.ln_stats_out else 0))
    return a + w + c + r + st


rows = collections.OrderedDict()
for i, (key, d, tile, split) in enumerate(seq):
    t_us = 1e3 * float(med[gpos[i]])
    batch = max(d.batch, 1)
    fl = 2.0 * d.M * d.N * (d.K + d.K2) * batch
    by = alg_bytes(d)
    floor_us = 1e6 * max(fl / MFMA_FLOOR, by / HBM_FLOOR)
    k = key + (tile, split)
    r = rows.setdefault(k, dict(n=0, t=0.0, fl=fl, by=by, floor=floor_us, tile=tile, split=split, desc=describe(key[:-1]) + (' +GN' if key[-1] else ''), key=key))
    r['n'] += 1
    r['t'] += t_us
tab = sorted(rows.values(), key=lambda r: -(r['t'] - r['n'] * r['floor']))
tot_t = sum(r['t'] for r in tab); tot_floor = sum(r['n'] * r['floor'] for r in tab); tot_fl = sum(r['n'] * r['fl'] for r in tab)
print(f'# GEMM-family loss table of ONE CFG UNet forward (SD1.5, CFG batch 16, 64x64 latents), launch-plan replay, {R} replays, every launch bracketed;')
print(f'# per launch position: median bracket - empty bracket ({1e3 * empty_ms:.2f} us).  floor = max(FLOPs / 1.2 PFLOP/s, algorithmic bytes / 6.3 TB/s).')
print(f'# forward wall time without brackets {wall_ms:.3f} ms; bracketed families per forward: ' + ', '.join(f'{k} {v:.3f} ms' for k, v in fam_ms.items()) +
      f'; sum {sum(fam_ms.values()):.3f} ms; gaps = wall - sum = {wall_ms - sum(fam_ms.values()):.3f} ms over {n_all} launches')
print(f'# table total {tot_t / 1e3:.3f} ms = the gemm family above ({fam_ms["gemm"]:.3f} ms): ratio {tot_t / 1e3 / fam_ms["gemm"]:.4f};  floor total {tot_floor / 1e3:.3f} ms;  '
      f'loss total {(tot_t - tot_floor) / 1e3:.3f} ms;  family rate {tot_fl / (tot_t * 1e-6) / 1e12:.0f} TFLOP/s executed')
print(f'{"#":>2} {"launch":<44} {"M":>6} {"N":>5} {"K(+K2)":>11} {"tile":>4} {"sk":>2} {"n":>3} {"us":>7} {"floor":>6} {"bound":>5} {"TF/s":>5} {"loss us":>7} {"x n":>8} {"cum %":>6}')
cum = 0.0
for i, r in enumerate(tab):
    key = r['key']
    each = r['t'] / r['n']
    loss = each - r['floor']
    cum += loss * r['n']
    bound = 'mfma' if r['fl'] / MFMA_FLOOR >= r['by'] / HBM_FLOOR else 'hbm'
    kk = f"{key[2]}" + (f"+{key[3]}" if key[3] else '')
    print(f'{i + 1:>2} {r["desc"][:44]:<44} {key[0] * max(key[13], 1):>6} {key[1]:>5} {kk:>11} {r["tile"]:>4} {r["split"]:>2} {r["n"]:>3} {each:>7.1f} {r["floor"]:>6.1f} {bound:>5} '
          f'{r["fl"] / (each * 1e-6) / 1e12:>5.0f} {loss:>7.1f} {loss * r["n"]:>8.1f} {100 * cum / max(tot_t - tot_floor, 1e-9):>6.1f}')
