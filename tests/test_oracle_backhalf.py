'''Pins the oracle's restatements of the reference's OWN back-half code -- SimpleGuide.noise_pred
(pipeline/guide.py:46-64) and CompositeGuide.noise_pred (composition/guide.py:56-139) -- against
goldens captured from the reference itself running with a recording stub UNet
(tests/golden/make_backhalf_goldens.py).  The UNet / VAE / scheduler arithmetic behind the stub
stays unpinned (diffusers 0.3.0 is absent).  CPU only.'''
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pipeline_ref, sched_ref

L, D = 77, 64


def stub_embed(prompt: str) -> torch.Tensor:
    '''Same seeded construction as tests/golden/make_backhalf_goldens.py::stub_embed.'''
    g = torch.Generator().manual_seed(zlib.crc32(prompt.encode()) & 0x7fffffff)
    return torch.randn((1, L, D), generator=g)


@pytest.fixture(scope='module')
def bh():
    return np.load(os.path.join(GOLDEN, 'backhalf_goldens.npz'))


class Replay():
    '''Replays the stub UNet's recorded output and checks it is called like the reference called it.'''

    def __init__(self, g, key):
        self.lat, self.ctx, self.out = (torch.from_numpy(g[f'{key}/{n}']) for n in
                                        ('unet_latents', 'unet_ctx', 'unet_out'))
        self.calls = 0

    def __call__(self, latents, t, ctx):
        assert torch.equal(latents, self.lat), 'latents stacked differently from the reference'
        assert torch.equal(ctx, self.ctx), 'embedding stack ordered differently from the reference'
        self.calls += 1
        return self.out


@pytest.mark.parametrize('name', ['cfg_b2', 'nocfg_b2', 'cfg_b1'])
def test_simple_guide_noise_pred_matches_reference(bh, name):
    key = f'simple/{name}'
    prompts = [str(p) for p in bh[key + '/prompts']]
    embeds = torch.cat([stub_embed(p) for p in prompts])
    rep = Replay(bh, key)
    got = pipeline_ref.noise_pred(None, None, torch.from_numpy(bh[key + '/latents']), int(bh[key + '/step'][0]),
                                  embeds, stub_embed(''), float(bh[key + '/guidance'][0]), unet_fn=rep)
    assert rep.calls == 1
    assert torch.equal(got, torch.from_numpy(bh[key + '/noise_pred'])), name


def composite_entities(bh, key):
    ents = []
    for (ox, oy, sw, sh), blend, p in zip(bh[key + '/entities'], bh[key + '/blend'], bh[key + '/entity_prompts']):
        ents.append((stub_embed(str(p)), (int(ox) // 8, int(oy) // 8), (int(sw) // 8, int(sh) // 8), float(blend)))
    return ents


def test_composite_guide_matches_reference(bh):
    for name in (str(n) for n in bh['composite/names']):
        key = f'composite/{name}'
        rep = Replay(bh, key)
        fn = lambda lat, emb: rep(lat, 500, emb)
        got = sched_ref.composite_noise_pred(fn, torch.from_numpy(bh[key + '/latents']), stub_embed(''),
                                             stub_embed(str(bh['composite/background_prompt'])),
                                             composite_entities(bh, key), float(bh[key + '/guidance'][0]))
        assert rep.calls == 1
        assert torch.equal(got, torch.from_numpy(bh[key + '/noise_pred'])), name
    # the negative-offset case really wraps: some box is blended away from the origin
    u = bh['composite/negative/unet_out']
    plain = u[0] + 8.0 * (u[1] - u[0])
    changed = np.abs(bh['composite/negative/noise_pred'][0] - plain).max(axis=0) > 1e-6
    assert changed[10:13, 8:12].all() and not changed[:, 12:].any() and not changed[13:].any()


def test_runner_compose_row_parsing():
    '''Runner.compose's forgiving row parser (utils.py:188-201) without a device: rows that do
    not parse are skipped, empty prompts dropped, ints / floats coerced.'''
    from flexdiffuse_amd import utils
    seen = {}

    class FakeRunner(utils.Runner):
        def __init__(self):
            self.encoder, self.pipe = None, type('P', (), {'unet': None})()
            self.generator = torch.Generator('cpu')
            self.eta = 0.0

        def _run(self, batches, guide, init_image, init_size, strength, debug):
            seen.update(batches=batches, guide=guide, init_size=init_size, strength=strength)
            return [], None

    orig = utils.CompositeGuide
    utils.CompositeGuide = lambda enc, unet, g, schema, steps: ('guide', g, schema, steps)
    try:
        r = FakeRunner()
        r.compose('a forest', [[' a deer ', '0', 16.0, 64, '48', '0.8'], ['', 0, 0, 8, 8, 0.5],
                               ['broken', 'x', 0, 8, 8, 0.5], ['short row'], ['a bird', 64, 0, 64, 64, 1]],
                  seed=5, batches=2, steps=7)
    finally:
        utils.CompositeGuide = orig
    _, gscale, schema, steps = seen['guide']
    assert (gscale, steps, seen['batches'], seen['strength'], seen['init_size']) == (8.0, 7, 2, 0.7, (512, 512))
    assert [e.prompt for e in schema.entities] == ['a deer', 'a bird']
    assert schema.entities[0].offset == (0, 16) and schema.entities[0].size == (64, 48)
    assert schema.entities[1].blend == 1.0 and schema.style_blend == (0.0, 1.0)
    assert schema.background_prompt == 'a forest'
