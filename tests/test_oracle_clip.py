'''Pins oracle/clip_ref.py + oracle/guide_ref.py against golden vectors captured from
the reference's encode/clip.py and guidance.Guide driving a tiny seeded transformers
CLIPModel (tests/golden/make_clip_goldens.py).  CPU only.'''
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flexdiffuse_amd import weights as W
from flexdiffuse_amd.tokenizer import SyntheticTokenizer
from oracle import clip_ref, guide_ref


def synth_image(seed, w, h):
    '''Same construction as tests/golden/make_clip_goldens.py::synth_image.'''
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h + 8, w + 8, 3)).astype(np.float32)
    c = np.cumsum(np.cumsum(a, 0), 1)
    c = np.pad(c, ((1, 0), (1, 0), (0, 0)))
    blur = (c[8:, 8:] - c[:-8, 8:] - c[8:, :-8] + c[:-8, :-8]) / 64.0
    blur = blur[:h, :w]
    blur = (blur - blur.min()) / (blur.max() - blur.min()) * 255.0
    return Image.fromarray(blur.astype(np.uint8), 'RGB')


@pytest.fixture(scope='module')
def cg():
    return np.load(os.path.join(GOLDEN, 'clip_goldens.npz'))


@pytest.fixture(scope='module')
def mini_sd(cg):
    return {k[3:]: torch.from_numpy(cg[k].astype(np.float32)) for k in cg.files
            if k.startswith('sd/')}


def test_text_tower(cg, mini_sd):
    tok = SyntheticTokenizer(vocab_size=W.MINI_CLIP.text.vocab_size)
    for i, p in enumerate(cg['prompts']):
        ids = tok(str(p)).input_ids
        assert np.array_equal(ids.numpy(), cg[f'prompt{i}/ids'])
        got = clip_ref.text_hidden(mini_sd, W.MINI_CLIP, ids).numpy()
        assert np.max(np.abs(got - cg[f'prompt{i}/hidden'])) < 2e-5, i
    ids = tok([str(p) for p in cg['prompts'][:2]]).input_ids
    got = clip_ref.text_hidden(mini_sd, W.MINI_CLIP, ids).numpy()
    assert np.max(np.abs(got - cg['prompt_batch/hidden'])) < 2e-5


def test_preprocess_and_vision_tower(cg, mini_sd):
    for i, (w, h) in enumerate(cg['image_sizes']):
        img = synth_image(20 + i, int(w), int(h))
        pre = clip_ref.preprocess(img)
        assert tuple(pre.shape) == tuple(cg[f'image{i}/pre_shape'])
        s = np.array([pre.double().sum().item(), pre.double().abs().sum().item()])
        assert np.allclose(s, cg[f'image{i}/pre_sum'], rtol=1e-9)
        if f'image{i}/pixels' in cg.files:
            px = clip_ref.clip_pixels(pre)
            assert np.max(np.abs(px.numpy() - cg[f'image{i}/pixels'].astype(np.float32))) < 4e-3
            st = np.array([px.double().sum().item(), px.double().abs().sum().item()])
            assert np.allclose(st, cg[f'image{i}/pixels_stat'], rtol=1e-6)
            tokens = clip_ref.image_tokens(mini_sd, W.MINI_CLIP, px).numpy()
            assert tokens.shape == (1, 257, W.MINI_CLIP.projection_dim)
            assert np.max(np.abs(tokens - cg[f'image{i}/tokens'])) < 5e-5, i


def test_sd_size_table():
    '''SURVEY App. A.6.'''
    table = {(512, 512): (512, 512), (900, 600): (512, 320), (600, 900): (320, 512),
             (512, 704): (320, 512), (896, 1024): (448, 512), (1000, 999): (512, 448),
             (100, 100): (512, 512)}
    for (w, h), want in table.items():
        assert clip_ref.sd_size(w, h) == want


def test_guide_embeds_branches(cg, mini_sd):
    tok = SyntheticTokenizer(vocab_size=W.MINI_CLIP.text.vocab_size)
    g = guide_ref.GuideRef(mini_sd, W.MINI_CLIP, tok)
    img = synth_image(20, 512, 512)
    p = [str(x) for x in cg['prompts']]
    tol = 5e-5
    assert np.max(np.abs(g.placeholder.numpy() - cg['guide/placeholder'])) < tol
    cases = {
        'text_only': dict(prompt=p[0]),
        'text_batch': dict(prompt=p[:2]),
        'image_linear': dict(prompt=p[0], guide=img, guide_threshold_mult=0.0,
                             guide_clustered=0.0, guide_linear=(0.0, 0.5),
                             guide_max_guidance=0.5),
        'image_thr': dict(prompt=p[1], guide=img, guide_threshold_mult=0.25,
                          guide_threshold_floor=0.05, guide_clustered=0.0,
                          guide_linear=(0.0, 0.0), guide_max_guidance=0.35, guide_header_max=0.0),
        'text_guide': dict(prompt=p[0], guide=p[1], guide_clustered=0.0),
        'pure_image': dict(guide=img),
        'pure_text_guide': dict(guide=p[1]),
        'concepts': dict(prompt=p[0], guide=img, mapping_concepts='turtle photo',
                         guide_clustered=0.0),
    }
    for name, kw in cases.items():
        got = g.embeds(**kw).numpy()
        want = cg['guide/' + name]
        assert got.shape == want.shape, name
        assert np.max(np.abs(got - want)) < tol, (name, np.max(np.abs(got - want)))
    with pytest.raises(ValueError):
        g.embeds(prompt=3)
    with pytest.raises(ValueError):
        g.embeds(prompt='', guide=None)
