import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def guidance_goldens():
    return np.load(os.path.join(GOLDEN, 'guidance_goldens.npz'))


def golden_scene(seed: int, n_alt: int, dim: int, planted: int, noise: float = 0.15):
    '''Same seeded construction as tests/golden/make_guidance_goldens.py::scene.'''
    L = 77
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


def golden_scene_soft(seed: int, n_alt: int, dim: int, planted: int):
    '''Same seeded construction as tests/golden/make_guidance_goldens.py::scene_soft
    (non-adjacent, non-saturating planted matches: clustered guidance yields weights).'''
    L = 77
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


SMALL_SCENES = ['s0_257x64', 's1_257x64', 's2_77x64', 's3_40x64', 's4_257x64_sharp']
BIG_SCENES = ['b0_257x768', 'b1_257x1024', 'b2_77x768']
SOFT_SCENES = ['c0_257x768', 'c1_257x1024', 'c2_77x768']


def load_scene(g, name):
    '''(alt, txt) for a golden scene: stored for the small ones, regenerated from the
    seed (and checked against the stored sha256) for the large ones.'''
    import hashlib
    if name + '/alt' in g.files:
        return g[name + '/alt'], g[name + '/txt']
    spec = [int(v) for v in g[name + '/spec']]
    seed, n, d, p = spec[:4]
    alt, txt = golden_scene_soft(seed, n, d, p) if len(spec) > 4 and spec[4] == 1 \
        else golden_scene(seed, n, d, p)
    sha = hashlib.sha256(alt.tobytes() + txt.tobytes()).digest()
    assert sha == g[name + '/sha'].tobytes(), 'seeded scene regeneration drifted'
    return alt, txt
