'''GPU: `flexdiffuse_amd.FlexPipeline` (constructor + `__call__`) replayed against the call traces
the reference's OWN pipeline/flex.py:46-83,112-124,126-310 produced with the same recording stubs
(tests/golden/flexcall_goldens.npz; generator tests/golden/make_flexcall_goldens.py).

The stubs stand where the reference receives diffusers' vae / scheduler / unet and a guide; the
product pipeline runs its generic protocol path (`guide.noise_pred` + `scheduler.step`,
`vae.encode(...).latent_dist.sample`, `vae.decode(...).sample`) with its own device ops around them
(fd_axpby_f32 scalings, the layout kernel's affine + clamp), tensors on the HIP device, noise from
the CPU generator.  Required: the same call ORDER (set_timesteps twice for txt2img, encode ->
sample -> add_noise for img2img, noise_pred / step pairs), the same timesteps / indices / add_noise
levels / eta forwarding, the same tensors within fp32 rounding, the same returned images (float
NHWC or uint8 PIL), the same tuple / record shape, the same constructor warning and config
rewrite.'''
import os

import numpy as np
import pytest
import torch

import flexcall_stubs as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'flexcall_goldens.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def _bases():
    from flexdiffuse_amd.scheduler import DDIMScheduler, LMSDiscreteScheduler, PNDMScheduler
    return {'ddim': DDIMScheduler, 'pndm': PNDMScheduler, 'lms': LMSDiscreteScheduler}


def _factory(vae, clip, tokenizer, unet, scheduler):
    from flexdiffuse_amd import hip
    from flexdiffuse_amd.pipeline.flex import FlexPipeline
    hip.lib()
    return FlexPipeline(vae, clip, tokenizer, unet, scheduler).to('cuda')


@pytest.mark.parametrize('name', list(S.CASES))
def test_product_call_trace_equals_reference(gold, name):
    trace, images, flags, ctor_warnings = S.run_case(name, _factory, _bases(), device='cuda')
    want = S.Trace.from_npz(f'{name}/trace', gold)
    S.assert_same_trace(trace, want, name)
    assert str(images.dtype) == str(gold[f'{name}/images_dtype'])
    head = f'{name}/images'
    want_img = {k[len(head):]: gold[k] for k in gold.files if k.startswith(head) and not k.endswith('_dtype')}
    if images.dtype == np.uint8:       # PIL output: a rounding flip of one grey level is fp32 noise
        got = S.digest(images.astype(np.float32))
        want_f = {k: (v.astype(np.float32) if k in ('', '__sub') else v) for k, v in want_img.items()}
        for k in want_f:
            if k in ('', '__sub'):
                assert np.abs(got[k] - want_f[k]).max() <= 1.0, name
                assert (got[k] != want_f[k]).mean() < 1e-3, name
            elif k == '__shape':
                assert np.array_equal(got[k], want_f[k])
    else:
        S.assert_same(S.digest(images), want_img, f'{name}.images')
    assert [str(w) for w in gold[f'{name}/ctor_warnings'] if str(w)] == ctor_warnings
    if flags is not None:
        assert flags == [bool(f) for f in gold[f'{name}/flags']]


@pytest.mark.parametrize('bad', (-0.1, 1.5))
def test_strength_valueerror_text(gold, bad):
    S.CASES['_bad'] = dict(kind='ddim', steps_offset=None, B=1, steps=4, init=None,
                           kw=dict(strength=bad, init_size=(64, 64)))
    try:
        with pytest.raises(ValueError) as ei:
            S.run_case('_bad', _factory, _bases(), device='cuda')
    finally:
        S.CASES.pop('_bad')
    assert str(ei.value) == str(gold[f'valueerror/{bad}'])
