'''Pins oracle/guidance_ref.py against golden vectors captured from the reference's
own guidance.py (tests/golden/make_guidance_goldens.py).  CPU only.'''
import hashlib
import itertools

import numpy as np
import pytest
import torch

from conftest import BIG_SCENES, SMALL_SCENES, SOFT_SCENES, load_scene
from oracle import guidance_ref as G


@pytest.mark.parametrize('name', SMALL_SCENES + BIG_SCENES + SOFT_SCENES)
def test_map_emb_matches_reference(guidance_goldens, name):
    g = guidance_goldens
    alt, txt = load_scene(g, name)
    sim = G.similarity(alt, txt)
    for mode, reuse in itertools.product((0, 1, 2), (True, False)):
        want = g[f'{name}/map_m{mode}_r{int(reuse)}']
        got = G.assign(sim, txt.shape[1], reuse, mode)
        assert np.array_equal(got[:, 0], want[:, 0]), (name, mode, reuse)
        # same machine & same per-row mat-vec order -> bit-exact; 1e-5 is the parity bar
        assert np.max(np.abs(got[:, 1] - want[:, 1])) <= 1e-5


@pytest.mark.parametrize('name', SMALL_SCENES + BIG_SCENES + SOFT_SCENES)
def test_tween_matches_reference(guidance_goldens, name):
    g = guidance_goldens
    alt, txt = load_scene(g, name)
    names = [str(n) for n in g['tween_sets/names']]
    for tname, vals in zip(names, g['tween_sets/values']):
        fl, mu, l0, l1, cl, mg, hm, mode, reuse = (float(v) for v in vals)
        key = f'{name}/tween_{tname}'
        kw = dict(threshold=(fl, mu), linear=(l0, l1), clustered=cl, max_guidance=mg,
                  header_max=hm, order=int(mode), reuse=bool(reuse))
        if key + '/zerodiv' in g.files:
            with pytest.raises(ZeroDivisionError):
                G.tween(txt, alt, **kw)
            continue
        out, w, _ = G.tween(txt, alt, **kw)
        assert np.max(np.abs(w.numpy() - g[key + '/weights'])) <= 1e-6, key
        if key + '/out' in g.files:
            assert np.max(np.abs(out.numpy() - g[key + '/out'])) <= 1e-6, key
        else:
            assert np.max(np.abs(out.numpy()[0, :, :8] - g[key + '/out_head'])) <= 1e-6
            sha = hashlib.sha256(out.numpy().tobytes()).digest()
            assert sha == g[key + '/out_sha'].tobytes(), key + ' (not bit-exact)'


def test_soft_scenes_exercise_clustered_guidance_at_real_widths(guidance_goldens):
    '''The D = 768 / 1024 scenes with non-adjacent, non-saturating matches must yield clustered
    WEIGHTS (not the ZeroDivisionError record the saturated scenes give) for every clustered
    parameter set, so the clustered / threshold path is value-pinned at the real widths.'''
    g = guidance_goldens
    names = [str(n) for n in g['tween_sets/names']]
    clustered_sets = [n for n, v in zip(names, g['tween_sets/values']) if v[4] != 0]
    assert len(clustered_sets) >= 6
    for scene in SOFT_SCENES:
        have = [n for n in clustered_sets if f'{scene}/tween_{n}/weights' in g.files]
        assert len(have) >= 3, (scene, have)
        s = g[f'{scene}/map_m1_r1'][:76, 1]
        assert 0.0 < s.max() < 1.0 and len(np.unique(np.round(s, 4))) > 60     # not saturated
        w = g[f'{scene}/tween_c3_clust_thr/weights']
        assert 0 < (w > 0).sum() < 77 and len(np.unique(w)) >= 4               # peaks, ramps and valleys


def test_clustered_and_blend_kats(guidance_goldens):
    g = guidance_goldens
    for s, thr, gain, code, w in zip(g['clustered/s'], g['clustered/thr'], g['clustered/gain'],
                                     g['clustered/code'], g['clustered/w']):
        m = np.zeros((77, 2))
        m[:, 1] = s
        if code == 2:
            with pytest.raises(ZeroDivisionError):
                G.clustered_weights(m, float(thr), float(gain))
            continue
        got = G.clustered_weights(m, float(thr), float(gain))
        if code == 1:
            assert got is None
        else:
            assert np.array_equal(got.numpy(), w)
    for a, b, r in zip(g['blend/a'], g['blend/b'], g['blend/r']):
        got = G.blend_weights(torch.from_numpy(a), torch.from_numpy(b)).numpy()
        assert np.array_equal(got, r)


def test_clustered_closed_form_kats():
    '''SURVEY App. A.2 known answers.'''
    m = np.zeros((77, 2)); m[10, 1] = 0.9
    w = G.clustered_weights(m, 0.5, 1.0).numpy()
    assert np.allclose(w[:11], np.arange(11) / 10.0, atol=1e-7)
    assert w[76] == 0 and abs(w[11] - (1 - 1 / 66)) < 1e-7
    m[20, 1] = 0.9
    w = G.clustered_weights(m, 0.5, 1.0).numpy()
    assert w[15] == 0 and np.allclose(w[11:15], [.8, .6, .4, .2], atol=1e-6)


@pytest.mark.parametrize('k', [0, 1, 2])
def test_concept_mapper_matches_reference(guidance_goldens, k):
    g = guidance_goldens
    img, concept, base = (g[f'concept{k}/{n}'] for n in ('img', 'concept', 'base'))
    plain = G.concept_override(img, concept, base)
    assert np.array_equal(plain.numpy(), g[f'concept{k}/out_plain'])
    over = G.concept_override(img, concept, base, out=g[f'concept{k}/tweened'])
    assert np.array_equal(over.numpy(), g[f'concept{k}/out'])
    assert not np.array_equal(g[f'concept{k}/out'], g[f'concept{k}/tweened']), 'vacuous'
