'''Golden-vector generator for the guidance stage -- runs in the BUILD CONTAINER only.

Imports the reference's own `guidance.py` (via tests/golden/_ref_loader.py), feeds
it seeded inputs and stores inputs + outputs as data in tests/golden/guidance_*.npz.
Usage:  python tests/golden/make_guidance_goldens.py
'''
import contextlib
import hashlib
import io
import itertools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

L = 77


def scene(seed: int, n_alt: int, dim: int, planted: int, noise: float = 0.15):
    '''Seeded (alt, txt) pair; `planted` text tokens are noisy copies of guide tokens
    so that high similarities (peaks, threshold hits) occur.'''
    rng = np.random.default_rng(seed)
    alt = rng.standard_normal((1, n_alt, dim)).astype(np.float32)
    txt = rng.standard_normal((1, L, dim)).astype(np.float32)
    if planted:
        tj = rng.choice(np.arange(1, L), size=planted, replace=False)
        ai = rng.choice(n_alt, size=planted, replace=True)
        for j, i in zip(tj, ai):
            txt[0, j] = alt[0, i] + noise * rng.standard_normal(dim).astype(np.float32) \
                * (0.2 + 2.0 * rng.random())
    return alt, txt


def scene_soft(seed: int, n_alt: int, dim: int, planted: int):
    '''Like `scene`, but built so that CLUSTERED guidance has something to do at real embedding
    widths: planted text tokens sit at NON-ADJACENT positions (every third column at most) and
    are buried in enough noise (cosine 0.08 .. 0.15 to their guide token) that the 77-way softmax
    of 100 x cosine does not saturate -- similarities land strictly between 0 and 1, peaks are
    isolated, and `_clustered_guidance` returns weights instead of raising ZeroDivisionError
    (guidance.py:112, adjacent equal peaks) as it does on the saturated `scene`s.'''
    rng = np.random.default_rng(seed)
    alt = rng.standard_normal((1, n_alt, dim)).astype(np.float32)
    txt = rng.standard_normal((1, L, dim)).astype(np.float32)
    slots = np.arange(2, L - 1, 3)
    tj = rng.choice(slots, size=min(planted, len(slots)), replace=False)
    ai = rng.choice(n_alt, size=len(tj), replace=False)
    for j, i in zip(tj, ai):
        cos = 0.08 + 0.07 * rng.random()
        sigma = np.sqrt(1.0 / cos ** 2 - 1.0)
        txt[0, j] = alt[0, i] + sigma * rng.standard_normal(dim).astype(np.float32)
    return alt, txt


TWEEN_SETS = {
    # name: (floor, mult, lin0, lin1, clustered, max_guidance, header_max, mode, reuse)
    'defaults':       (0.5, 0.5, 0.0, 0.5, 0.5, 0.5, 0.15, 1, True),
    'c2_linear':      (0.5, 0.0, 0.0, 0.5, 0.0, 0.5, 0.15, 1, True),
    'c3_clust_thr':   (0.75, 0.25, 0.0, 0.0, 0.25, 0.35, 0.0, 1, True),
    'readme_tuned':   (0.75, 0.25, 0.0, 0.5, 0.25, 0.35, 0.0, 0, True),
    'neg_linear':     (0.5, 0.5, -0.3, 0.2, 0.5, 0.5, 0.15, 1, False),
    'neg_all':        (0.2, -0.4, -0.5, -0.1, -0.3, 0.5, 0.15, 0, False),
    'neg_mult':       (0.3, -0.5, 0.0, 0.5, 0.5, 0.5, 1.0, 1, True),
    'hdr_full':       (0.1, 1.0, 0.9, 1.0, 1.0, 1.0, 1.0, 2, True),
    'thr_only':       (0.6, 0.8, 0.0, 0.0, 0.0, 0.7, 0.15, 1, True),
    'clust_only':     (0.5, 0.0, 0.0, 0.0, 1.0, 1.0, 0.15, 0, False),
}


def main():
    guidance, _ = load_reference()
    quiet = contextlib.redirect_stdout(io.StringIO())
    out = {}
    # ---- G1: _map_emb ------------------------------------------------------------
    scenes = {
        's0_257x64': (100, 257, 64, 12),
        's1_257x64': (101, 257, 64, 0),
        's2_77x64': (102, 77, 64, 20),
        's3_40x64': (103, 40, 64, 8),
        's4_257x64_sharp': (104, 257, 64, 40),
    }
    big = {'b0_257x768': (200, 257, 768, 16), 'b1_257x1024': (201, 257, 1024, 16),
           'b2_77x768': (202, 77, 768, 10)}
    for name, (seed, n, d, p) in scenes.items():
        alt, txt = scene(seed, n, d, p, noise=0.05 if 'sharp' in name else 0.15)
        out[f'{name}/alt'] = alt
        out[f'{name}/txt'] = txt
    for name, (seed, n, d, p) in big.items():
        alt, txt = scene(seed, n, d, p)
        out[f'{name}/spec'] = np.array([seed, n, d, p], dtype=np.int64)
        out[f'{name}/sha'] = np.frombuffer(
            hashlib.sha256(alt.tobytes() + txt.tobytes()).digest(), dtype=np.uint8)
    # soft scenes (kind 1): non-adjacent, non-saturating matches at the real embedding widths
    soft = {'c0_257x768': (210, 257, 768, 14), 'c1_257x1024': (211, 257, 1024, 14),
            'c2_77x768': (212, 77, 768, 10)}
    for name, (seed, n, d, p) in soft.items():
        alt, txt = scene_soft(seed, n, d, p)
        out[f'{name}/spec'] = np.array([seed, n, d, p, 1], dtype=np.int64)
        out[f'{name}/sha'] = np.frombuffer(
            hashlib.sha256(alt.tobytes() + txt.tobytes()).digest(), dtype=np.uint8)
    allscenes = {**scenes, **big, **soft}
    n_clustered_ok = {}
    for name, (seed, n, d, p) in allscenes.items():
        if name in soft:
            alt, txt = scene_soft(seed, n, d, p)
        else:
            alt, txt = scene(seed, n, d, p, noise=0.05 if 'sharp' in name else 0.15)
        ta, tt = torch.from_numpy(alt), torch.from_numpy(txt)
        for mode, reuse in itertools.product((0, 1, 2), (True, False)):
            with quiet:
                m = guidance._map_emb(ta, tt, reuse, mode)
            out[f'{name}/map_m{mode}_r{int(reuse)}'] = m
        # ---- G2: Tweener.tween ---------------------------------------------------
        for tname, (fl, mu, l0, l1, cl, mg, hm, mode, reuse) in TWEEN_SETS.items():
            tw = guidance.Tweener((fl, mu), (l0, l1), cl, mg, hm, mode, reuse)
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    res = tw.tween(tt, ta)
            except ZeroDivisionError:
                out[f'{name}/tween_{tname}/zerodiv'] = np.array([1])
                continue
            # weights are only observable through the print: recompute them by the
            # reference's own helpers for the golden
            with quiet:
                m = guidance._map_emb(ta, tt, reuse, mode)
                w = torch.linspace(l0, l1, steps=L)
                if cl != 0:
                    cw = guidance._clustered_guidance(m, m[:, 1].mean(), cl)
                    if cw is not None:
                        w = guidance._blend_weights(w, cw)
                if mu != 0:
                    th = torch.ones_like(w) * mu
                    for j, (_, s) in enumerate(m):
                        if s < fl:
                            th[j] = 0
                    w = guidance._blend_weights(w, th)
                if hm < 1.0:
                    hw = w[0].item()
                    w[0] = min(hw, hm) if hw >= 0 else max(hw, -hm)
            key = f'{name}/tween_{tname}'
            out[key + '/weights'] = w.numpy()
            if cl != 0:
                n_clustered_ok[name] = n_clustered_ok.get(name, 0) + 1
            if d <= 64:
                out[key + '/out'] = res.numpy()
            else:
                out[key + '/out_sha'] = np.frombuffer(
                    hashlib.sha256(res.numpy().tobytes()).digest(), dtype=np.uint8)
                out[key + '/out_head'] = res.numpy()[0, :, :8].copy()
    for name in soft:
        assert n_clustered_ok.get(name, 0) >= 3, (name, n_clustered_ok)
        m = out[f'{name}/map_m1_r1'][:, 1]
        assert 0.0 < m[:76].max() < 1.0, (name, m.max())        # not saturated
    print('clustered tween records with weights per scene:', n_clustered_ok)
    out['tween_sets/names'] = np.array(list(TWEEN_SETS.keys()))
    out['tween_sets/values'] = np.array([[float(v) for v in vals]
                                         for vals in TWEEN_SETS.values()])
    # ---- G3: clustered / blend KATs ---------------------------------------------
    rng = np.random.default_rng(300)
    kat = []
    for k in range(24):
        m = np.zeros((L, 2))
        m[:76, 1] = rng.random(76).astype(np.float32)
        if k % 3 == 0:
            m[:76, 1] = (m[:76, 1] ** 4).astype(np.float32)
        m[:, 0] = rng.integers(0, 257, L)
        thr = float(m[:, 1].mean()) if k % 2 == 0 else float(rng.random())
        gain = float(rng.choice([1.0, 0.25, -0.5, 0.7]))
        try:
            with quiet:
                cw = guidance._clustered_guidance(m, thr, gain)
            code = 0 if cw is not None else 1
        except ZeroDivisionError:
            cw, code = None, 2
        kat.append((m[:, 1].copy(), thr, gain, code,
                    cw.numpy() if cw is not None else np.zeros(L, np.float32)))
    # explicit single/double peak KATs
    for peaks in ([10], [10, 20], [1, 75], [10, 11], [5, 40, 41 + 20]):
        m = np.zeros((L, 2))
        for p in peaks:
            m[p, 1] = 0.9
        try:
            with quiet:
                cw = guidance._clustered_guidance(m, 0.5, 1.0)
            code = 0 if cw is not None else 1
        except ZeroDivisionError:
            cw, code = None, 2
        kat.append((m[:, 1].copy(), 0.5, 1.0, code,
                    cw.numpy() if cw is not None else np.zeros(L, np.float32)))
    out['clustered/s'] = np.stack([k[0] for k in kat])
    out['clustered/thr'] = np.array([k[1] for k in kat])
    out['clustered/gain'] = np.array([k[2] for k in kat])
    out['clustered/code'] = np.array([k[3] for k in kat])
    out['clustered/w'] = np.stack([k[4] for k in kat])
    bl = []
    for sa, sb in itertools.product((1, -1, 0), (1, -1, 0)):
        a = (rng.random(L).astype(np.float32) * sa - (0.2 if sa < 0 else 0)).astype(np.float32)
        b = (rng.random(L).astype(np.float32) * sb - (0.2 if sb < 0 else 0)).astype(np.float32)
        r = guidance._blend_weights(torch.from_numpy(a), torch.from_numpy(b)).numpy()
        bl.append((a, b, r))
    out['blend/a'] = np.stack([b[0] for b in bl])
    out['blend/b'] = np.stack([b[1] for b in bl])
    out['blend/r'] = np.stack([b[2] for b in bl])
    # ---- G4: ConceptMapper -------------------------------------------------------
    for k, seed in enumerate((400, 401, 402)):
        rng = np.random.default_rng(seed)
        img = rng.standard_normal((1, 257, 64)).astype(np.float32)
        concept = rng.standard_normal((1, L, 64)).astype(np.float32)
        base = rng.standard_normal((1, L, 64)).astype(np.float32)
        # concept tokens 1..6 copy image tokens; base tokens copy concept tokens
        for c in range(1, 7):
            concept[0, c] = img[0, 10 * c + k] + 0.02 * rng.standard_normal(64).astype(np.float32)
        for j, c in ((3, 1), (9, 2), (20, 4), (33, 6), (50, 3)):
            base[0, j] = concept[0, c] + (0.01 + 0.05 * k) * rng.standard_normal(64).astype(np.float32)
        with quiet:
            cmapper = guidance.ConceptMapper(torch.from_numpy(img), torch.from_numpy(concept))
            tw = guidance.Tweener(clustered=0.0)
            tweened = tw.tween(torch.from_numpy(base), torch.from_numpy(img))
            res = cmapper.map(torch.from_numpy(base), tweened.clone())
            res_plain = cmapper.map(torch.from_numpy(base))
        out[f'concept{k}/img'] = img
        out[f'concept{k}/concept'] = concept
        out[f'concept{k}/base'] = base
        out[f'concept{k}/tweened'] = tweened.numpy()
        out[f'concept{k}/out'] = res.numpy()
        out[f'concept{k}/out_plain'] = res_plain.numpy()
    path = os.path.join(HERE, 'guidance_goldens.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
