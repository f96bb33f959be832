'''Host-side logic of the product package against the oracle / known answers.  CPU only
(no kernel launches).'''
import numpy as np
import pytest
import torch

from flexdiffuse_amd import ops, weights as W
from flexdiffuse_amd.encode import clip as eclip
from flexdiffuse_amd.scheduler import DDIMScheduler
from flexdiffuse_amd.tokenizer import SyntheticTokenizer
from oracle import clip_ref, ddim_ref


def test_scheduler_tables_match_oracle_bit_exact():
    s = DDIMScheduler()
    assert np.array_equal(s.alphas_cumprod, ddim_ref.alphas_cumprod().numpy())
    for n, off in ((50, 0), (50, 1), (10, 0), (30, 0)):
        s = DDIMScheduler(steps_offset=off)
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps, ddim_ref.timesteps(n, steps_offset=off))
    s = DDIMScheduler()
    s.set_timesteps(50)
    c1, c2, c3, c4, sig = s.step_coefficients(500)
    acp = ddim_ref.alphas_cumprod()
    assert c1 == np.sqrt(1 - acp[500].numpy()) and c3 == np.sqrt(acp[480].numpy()) and sig == 0
    c = s.step_coefficients(0)
    assert c[2] == np.sqrt(acp[0].numpy())          # set_alpha_to_one=False
    assert s.set_format('pt') is s and s.config.get('steps_offset', 0) == 0


def test_img2img_schedule_arithmetic():
    '''steps 50 / strength 0.6 -> noise at t=580, 30 steps remain.'''
    s = DDIMScheduler()
    s.set_timesteps(50)
    init_timestep = min(int(50 * 0.6) + 0, 50)
    assert int(s.timesteps[-init_timestep]) == 580
    assert max(50 - init_timestep + 0, 0) == 20 and len(s.timesteps[20:]) == 30


def test_tokenizer_surface():
    tok = SyntheticTokenizer()
    ids = tok(['a photo of a turtle', ''], padding='max_length', max_length=77, truncation=True,
              return_tensors='pt').input_ids
    assert ids.shape == (2, 77) and ids.dtype == torch.long
    assert ids[0, 0] == 49406 and ids[0, 6] == 49407 and (ids[1, 1:] == 49407).all()
    long = tok(' '.join(['w'] * 200)).input_ids
    assert long.shape == (1, 77) and long[0, -1] == 49407
    assert torch.equal(tok('A Photo').input_ids, tok('a photo').input_ids)


def test_preprocess_matches_oracle():
    from test_oracle_clip import synth_image
    for (w, h) in ((512, 512), (900, 600), (512, 704), (100, 100)):
        img = synth_image(3, w, h)
        a, b = eclip.preprocess(img), clip_ref.preprocess(img)
        assert torch.equal(a, b)
        assert torch.equal(eclip.clip_pixels(a), clip_ref.clip_pixels(b))
    assert eclip.sd_size(1000, 999) == (512, 448)


def test_geglu_weight_interleave():
    C = 32
    w, b = torch.randn(8 * C, C), torch.randn(8 * C)
    lw = ops.prep_geglu(w, b, 'cpu')
    wi = lw.w.float()
    assert torch.allclose(wi[0:16], w[0:16].half().float())               # value rows 0..15
    assert torch.allclose(wi[16:32], w[4 * C:4 * C + 16].half().float())  # gate rows 0..15
    assert torch.allclose(wi[32:48], w[16:32].half().float())
    assert torch.equal(lw.bias[16:32], b[4 * C:4 * C + 16])


def test_conv_weight_layouts():
    w = torch.randn(8, 64, 3, 3)
    cw = ops.prep_conv(w, torch.randn(8), 'cpu')
    assert not cw.im2col and cw.w.shape == (8, 9 * 64)
    assert torch.equal(cw.w[3, 64:128], w[3, :, 0, 1].half())            # tap (0,1), all channels
    cw = ops.prep_conv(torch.randn(16, 4, 3, 3), None, 'cpu', cin_pad=8)
    assert cw.im2col and cw.kpad == 128 and cw.cin == 8 and cw.bias is None
    assert float(cw.w[:, 72:].abs().max()) == 0.0


def test_reference_export_names():
    import flexdiffuse_amd as F
    for name in ('CLIPEncoder', 'GUIDE_ORDER_TEXT', 'GUIDE_ORDER_ALIGN', 'Guide', 'preprocess',
                 'FlexPipeline'):
        assert hasattr(F, name)
    assert (F.GUIDE_ORDER_TEXT, F.GUIDE_ORDER_ALIGN, F.GUIDE_ORDER_DIRECT) == (0, 1, 2)
    import inspect
    sig = inspect.signature(F.Guide.embeds)
    assert list(sig.parameters)[1:] == ['prompt', 'guide', 'mapping_concepts', 'guide_threshold_mult',
                                        'guide_threshold_floor', 'guide_clustered', 'guide_linear',
                                        'guide_max_guidance', 'guide_header_max', 'guide_mode',
                                        'guide_reuse']
    d = {k: v.default for k, v in sig.parameters.items() if k != 'self'}
    assert d['guide_linear'] == (0.0, 0.5) and d['guide_header_max'] == 0.15 and d['guide_mode'] == 1
    sig = inspect.signature(F.FlexPipeline.__call__)
    assert list(sig.parameters)[1:10] == ['guide', 'init_image', 'init_size', 'strength', 'eta',
                                          'generator', 'output_type', 'return_dict', 'debug']
