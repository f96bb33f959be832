'''Performance guard of fd_gemm_f16's tile / split-K rule: correctness off-preset is covered by
tests/test_gpu_kernels.py::test_gemm_rule_cascade_on_shapes_outside_the_presets, but a regression INSIDE a rule only showed up as a bench
delta.  Here every unique GEMM / convolution launch of one full-size SD1.5 CFG forward (batch 8 -> CFG batch 16, 64x64 latents: the
bench's workload) is recorded by value (tools/gemm_recorder.py) and re-issued (a) through the rule and (b) with a small set of forced
(tile, split_k) candidates; per launch and in total the rule must not be slower than the best candidate by more than the stated margin.
Interleaved, best of three rounds per arm (clock drift between arms of one launch is the noise to beat).  Needs an MI355X.'''
import ctypes
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CANDIDATE_TILES = (9, 12, 13, 14, 15, 16, 20, 30, 31, 32, 33)
CANDIDATE_SPLITS = (1, 2, 4, 8)


def _time(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _guard(rec, gemm_recorder, min_launches):
    from flexdiffuse_amd import hip
    st = hip.stream()
    lib = hip.lib()
    assert len(rec) >= min_launches, len(rec)
    rows = []
    for key, (d, cnt) in rec.items():
        k = dict(zip(gemm_recorder.KEY_FIELDS, key))

        def run(tile, sk):
            d.tile, d.split_k = tile, sk
            return lib.fd_gemm_f16(ctypes.byref(d), st)
        assert run(0, 0) == 0
        t_auto = _time(lambda: run(0, 0), 6)
        if t_auto * cnt < 0.02:                       # < 20 us per forward in total: below the timing noise that matters
            continue
        cands = []
        for tile in CANDIDATE_TILES:
            if (tile in (15, 31) and k['N'] % 256) or (tile in (16, 30, 32) and k['N'] % 320) or (tile >= 30 and (k['lno'] or k['trans'])):
                continue
            if k['act'] == 4 and tile not in (14, 15, 31):
                continue
            for sk in CANDIDATE_SPLITS:
                if sk > 1 and (k['act'] == 4 or k['batch'] > 1 or k['lnf'] or k['lno'] or k['K2'] or (k['K'] // 64) // sk < 8 or
                               sk * k['M'] * k['N'] * 4 > d.workspace_bytes):
                    continue
                if run(tile, sk) == 0:                # candidates the library refuses (FD_ESHAPE) are not candidates
                    cands.append((tile, sk))
        torch.cuda.synchronize()
        best = {}
        auto = []
        for _ in range(3):                            # interleaved rounds
            auto.append(_time(lambda: run(0, 0), 6))
            for c in cands:
                best[c] = min(best.get(c, 1e9), _time(lambda: run(*c), 4))
        d.tile, d.split_k = 0, 0
        t_auto = min(auto)
        if not best:
            continue
        c_best, t_best = min(best.items(), key=lambda kv: kv[1])
        rows.append((t_auto * cnt, t_best * cnt, t_auto, t_best, c_best, key, cnt))
    tot_auto, tot_best = sum(r[0] for r in rows), sum(min(r[0], r[1]) for r in rows)
    rows.sort(reverse=True)
    for ta, tb, a, b, c, key, cnt in rows[:12]:
        k = dict(zip(gemm_recorder.KEY_FIELDS, key))
        print(f"M={k['M']:6d} N={k['N']:5d} K={k['K']:5d} {gemm_recorder.describe(key):24s} x{cnt:2d}: rule {a * 1e3:7.1f} us, best forced {c} {b * 1e3:7.1f} us")
    print(f'rule {tot_auto:.2f} ms per forward over the guarded launches, best-per-launch of the candidate set {tot_best:.2f} ms')
    # in total the rule may trail the per-launch optimum of this candidate set by 3 % (noise averages out over ~60 launches); a
    # single launch that matters (>= 1 % of the total) by 10 % -- one launch pair moves by +-5 % between measurements on this pool, so
    # a suspect is re-timed (five more interleaved rounds) before it fails the test
    assert tot_auto <= 1.03 * tot_best, (tot_auto, tot_best)
    for ta, tb, a, b, c, key, cnt in rows:
        if ta >= 0.01 * tot_auto and a > 1.10 * b:
            d = rec[key][0]

            def run(tile, sk):
                d.tile, d.split_k = tile, sk
                return lib.fd_gemm_f16(ctypes.byref(d), st)
            a2, b2 = [], []
            for _ in range(5):
                a2.append(_time(lambda: run(0, 0), 8))
                b2.append(_time(lambda: run(*c), 8))
            d.tile, d.split_k = 0, 0
            a2, b2 = sorted(a2)[2], sorted(b2)[2]
            assert a2 <= 1.10 * b2, (gemm_recorder.describe(key), key[:3], 'rule', a2, 'best', c, b2)


def test_rule_is_within_margin_of_the_best_forced_tile():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gemm_recorder
    rec, keep = gemm_recorder.record('sd15', 64, 8, vae=False)
    _guard(rec, gemm_recorder, 30)
    del keep


def test_rule_on_the_vae_decoder_launches():
    '''the same guard over one full-size VAE decode (8 latents 64x64 -> 512x512: the bench's decode): widths 512 / 256 / 128'''
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gemm_recorder
    rec, keep = gemm_recorder.record('sd15', 64, 8, vae=True, unet=False)
    _guard(rec, gemm_recorder, 15)
    del keep
