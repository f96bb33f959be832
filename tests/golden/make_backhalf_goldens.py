'''Golden-vector generator for the parts of the BACK half the reference itself owns --
runs in the BUILD CONTAINER only.

diffusers 0.3.0 is not installed, so the UNet / VAE / scheduler arithmetic cannot be pinned
(oracle/__init__.py).  But two functions on the hot path are the reference's OWN pure-torch
code around the UNet call:

  * `SimpleGuide.noise_pred`           pipeline/guide.py:46-64   (CFG stack order + combine)
  * `CompositeGuide._guide_latents` /
    `.noise_pred`                      composition/guide.py:56-139 (region blend + CFG)

They are imported here from /root/reference behind an in-memory `diffusers.models` module that
only provides the `UNet2DConditionModel` NAME the two files import for type annotations, and
run with a recording stub in place of the UNet: the stub returns seeded random tensors and
records what it was called with.  Inputs, the stub's outputs and the reference's results are
stored as data in tests/golden/backhalf_goldens.npz; the tests then replay the same stub
outputs through oracle/pipeline_ref.noise_pred, oracle/sched_ref.composite_noise_pred and the
device SimpleGuide / CompositeGuide.

Usage:  python tests/golden/make_backhalf_goldens.py
'''
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

D = 64      # embedding width of the stub encoder
L = 77


def stub_embed(prompt: str) -> torch.Tensor:
    '''Deterministic (1, 77, D) embedding of a prompt string (seed = crc32 of the text).'''
    g = torch.Generator().manual_seed(zlib.crc32(prompt.encode()) & 0x7fffffff)
    return torch.randn((1, L, D), generator=g)


class StubEncoder():
    def prompt(self, p):
        if isinstance(p, str):
            return stub_embed(p)
        return torch.cat([stub_embed(s) for s in p])


class RecordingUNet():
    '''Stands where the reference passes diffusers' UNet2DConditionModel: returns a seeded random
    tensor shaped like its latent input and records the call.'''

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.calls = []

    def __call__(self, latents, step, encoder_hidden_states=None):
        out = torch.randn(latents.shape, generator=self.g)
        self.calls.append((latents.clone(), step, encoder_hidden_states.clone(), out.clone()))
        return types.SimpleNamespace(sample=out)


COMPOSITE_CASES = {
    # name: (guidance, H, W, [(prompt, (ox, oy) px, (sw, sh) px, blend)])
    'two_boxes': (8.0, 16, 16, [('a deer', (0, 16), (64, 48), 0.8), ('a red bird', (64, 0), (64, 64), 0.5)]),
    'overlap_no_cfg': (1.0, 16, 16, [('a deer', (16, 16), (64, 64), 0.8), ('a fox', (48, 32), (64, 64), 0.3)]),
    'clipped': (7.5, 16, 24, [('a tree', (160, 96), (64, 64), 0.6), ('a rock', (0, 0), (400, 400), 0.25)]),
    # negative offsets: Python slicing wraps them from the end of the axis (usually an empty box)
    'negative': (8.0, 16, 16, [('a cloud', (-16, 0), (32, 32), 0.9), ('a bird', (-64, -48), (32, 24), 0.7),
                               ('a hill', (8, -128), (64, 200), 0.5)]),
    'no_entities': (8.0, 8, 8, []),
}


def main():
    load_reference()                       # torchvision shim + sys.path for /root/reference
    dm = types.ModuleType('diffusers.models')
    dm.UNet2DConditionModel = object       # only the NAME is needed (type annotations)
    dpkg = types.ModuleType('diffusers')
    dpkg.models = dm
    sys.modules.update({'diffusers': dpkg, 'diffusers.models': dm})
    import pipeline.guide as pguide
    import composition.guide as cguide
    from composition.schema import EntitySchema, Schema

    out = {}
    enc = StubEncoder()
    # ---- SimpleGuide.noise_pred / PromptGuide ---------------------------------------------
    for name, guidance, prompts in (('cfg_b2', 8.0, ['a photo of a turtle', 'zeus, oil painting']),
                                    ('nocfg_b2', 1.0, ['a photo of a turtle', 'zeus, oil painting']),
                                    ('cfg_b1', 7.5, ['a castle'])):
        unet = RecordingUNet(1000 + len(out))
        guide = pguide.PromptGuide(enc, unet, guidance, 10, prompts)
        assert guide.batch_size == len(prompts)
        lat = torch.randn((len(prompts), 4, 8, 8), generator=torch.Generator().manual_seed(7))
        res = guide.noise_pred(lat, 500)
        (lat_in, step, ctx, unet_out), = unet.calls
        out[f'simple/{name}/guidance'] = np.array([guidance])
        out[f'simple/{name}/prompts'] = np.array(prompts)
        out[f'simple/{name}/latents'] = lat.numpy()
        out[f'simple/{name}/unet_latents'] = lat_in.numpy()
        out[f'simple/{name}/unet_ctx'] = ctx.numpy()
        out[f'simple/{name}/unet_out'] = unet_out.numpy()
        out[f'simple/{name}/noise_pred'] = res.numpy()
        out[f'simple/{name}/step'] = np.array([step])
    # ---- CompositeGuide.noise_pred ------------------------------------------------------------
    for name, (guidance, H, W, ents) in COMPOSITE_CASES.items():
        unet = RecordingUNet(2000 + len(out))
        schema = Schema('a forest at dawn', 'oil painting', 'photograph', (0.0, 1.0),
                        [EntitySchema(p, off, size, blend) for p, off, size, blend in ents])
        guide = cguide.CompositeGuide(enc, unet, guidance, schema, 10)
        lat = torch.randn((1, 4, H, W), generator=torch.Generator().manual_seed(11))
        res = guide.noise_pred(lat, 500)
        (lat_in, step, ctx, unet_out), = unet.calls
        out[f'composite/{name}/guidance'] = np.array([guidance])
        out[f'composite/{name}/entities'] = np.array(
            [[off[0], off[1], size[0], size[1]] for _, off, size, _ in ents], dtype=np.int64).reshape(-1, 4)
        out[f'composite/{name}/blend'] = np.array([b for *_, b in ents], dtype=np.float64)
        out[f'composite/{name}/entity_prompts'] = np.array([p for p, *_ in ents] or [''])
        out[f'composite/{name}/latents'] = lat.numpy()
        out[f'composite/{name}/unet_latents'] = lat_in.numpy()
        out[f'composite/{name}/unet_ctx'] = ctx.numpy()
        out[f'composite/{name}/unet_out'] = unet_out.numpy()
        out[f'composite/{name}/noise_pred'] = res.numpy()
    out['composite/names'] = np.array(list(COMPOSITE_CASES))
    out['composite/background_prompt'] = np.array('a forest at dawn')
    path = os.path.join(HERE, 'backhalf_goldens.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
