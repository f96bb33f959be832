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


def test_scheduler_config_and_pipeline_constructor_rewrite():
    '''pipeline/flex.py:57-70: a scheduler config that HAS a steps_offset other than 1 is rewritten to 1 with a
    DeprecationWarning; a config without the key (diffusers 0.3.0, and this package's steps_offset=0 schedulers) is left
    alone.  No device needed: the constructor launches nothing.'''
    import types
    import warnings
    from flexdiffuse_amd.pipeline.flex import FlexPipeline
    from flexdiffuse_amd.scheduler import PNDMScheduler
    s0, s1, s2 = DDIMScheduler(), DDIMScheduler(steps_offset=1), PNDMScheduler(steps_offset=2)
    assert not hasattr(s0.config, 'steps_offset') and s0.config.get('steps_offset', 0) == 0
    assert s1.config.steps_offset == 1 and s1.config['num_train_timesteps'] == 1000
    with pytest.raises(AttributeError):
        s0.config.no_such_key
    unet = types.SimpleNamespace(device=torch.device('cpu'))
    for sched, want_warn, want_off in ((s0, False, 0), (s1, False, 1), (s2, True, 1)):
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter('always')
            pipe = FlexPipeline(None, None, None, unet, sched)
        assert any(w.category is DeprecationWarning for w in wl) == want_warn
        assert pipe.scheduler.config.get('steps_offset', 0) == want_off
        pipe.scheduler.set_timesteps(10)
        assert int(pipe.scheduler.timesteps[-1]) == want_off
    assert pipe.pause_gc is False and pipe.use_plan is True and pipe.use_graph is True


def test_runner_constructor_refusals_and_from_directories(tmp_path):
    '''`Runner(local=False)` cannot be served (no hub access); directories must come in pairs; a checkpoint directory
    without a tokenizer is refused before any model is built.  CPU only: every refusal happens before the device is used.'''
    import os
    from safetensors.torch import save_file
    from flexdiffuse_amd import Runner, build
    with pytest.raises(RuntimeError):
        Runner(local=False, device='cuda')
    with pytest.raises(ValueError):
        Runner(True, 'cuda', sd_dir=str(tmp_path))
    sds = build.synthetic_state_dicts('mini_bpe', seed=1)
    for sub in ('unet', 'vae'):
        os.makedirs(tmp_path / 'sd' / sub)
        save_file({k: v.contiguous() for k, v in sds[sub].items()},
                  str(tmp_path / 'sd' / sub / 'diffusion_pytorch_model.safetensors'))
    os.makedirs(tmp_path / 'clip')
    save_file({k: v.contiguous() for k, v in sds['clip'].items()}, str(tmp_path / 'clip' / 'model.safetensors'))
    with pytest.raises(FileNotFoundError):
        build.from_directories(str(tmp_path / 'sd'), str(tmp_path / 'clip'), preset='mini_bpe', device='cuda')
    assert build.configs('mini_bpe')[2].text.vocab_size == 1024


def test_scheduler_comes_from_the_checkpoint_directory(tmp_path):
    '''the reference hands the checkpoint's own scheduler to its pipeline (utils.py:70 `sd.scheduler`; SD-v1-4 ships
    PNDM / PLMS), so `build.load_scheduler` maps scheduler/scheduler_config.json to this package's classes with the file's
    parameters, returns None without the folder and refuses classes it does not provide.'''
    import json
    import os
    from flexdiffuse_amd import build
    assert build.load_scheduler(str(tmp_path)) is None
    os.makedirs(tmp_path / 'scheduler')
    cfg = tmp_path / 'scheduler' / 'scheduler_config.json'
    cfg.write_text(json.dumps({'_class_name': 'PNDMScheduler', '_diffusers_version': '0.2.2', 'beta_end': 0.012,
                               'beta_schedule': 'scaled_linear', 'beta_start': 0.00085, 'num_train_timesteps': 1000,
                               'skip_prk_steps': True}))
    s = build.load_scheduler(str(tmp_path))
    assert type(s).__name__ == 'PNDMScheduler' and s.config['skip_prk_steps'] is True and s.config['beta_end'] == 0.012
    s.set_timesteps(50)
    assert len(s.timesteps) == 51                      # PLMS: one extra UNet evaluation
    cfg.write_text(json.dumps({'_class_name': 'DDIMScheduler', 'beta_start': 0.00085, 'beta_end': 0.012,
                               'beta_schedule': 'scaled_linear', 'num_train_timesteps': 1000, 'clip_sample': False,
                               'set_alpha_to_one': False, 'steps_offset': 1}))
    s = build.load_scheduler(str(tmp_path))
    assert type(s).__name__ == 'DDIMScheduler' and s.config['steps_offset'] == 1
    cfg.write_text(json.dumps({'_class_name': 'LMSDiscreteScheduler', 'beta_start': 0.00085, 'beta_end': 0.012,
                               'beta_schedule': 'scaled_linear', 'num_train_timesteps': 1000}))
    assert type(build.load_scheduler(str(tmp_path))).__name__ == 'LMSDiscreteScheduler'
    cfg.write_text(json.dumps({'_class_name': 'EulerDiscreteScheduler'}))
    with pytest.raises(NotImplementedError):
        build.load_scheduler(str(tmp_path))
    # PNDM / LMS have no v-prediction arithmetic -- a v-prediction checkpoint (sd21 / mini2 presets, or the file's own
    # prediction_type) with such a scheduler file is refused, never stepped as epsilon; DDIM carries the type through
    sd_pndm = {'_class_name': 'PNDMScheduler', 'beta_end': 0.012, 'beta_schedule': 'scaled_linear', 'beta_start': 0.00085,
               'num_train_timesteps': 1000, 'skip_prk_steps': True}
    cfg.write_text(json.dumps(sd_pndm))
    with pytest.raises(NotImplementedError, match='v_prediction'):
        build.load_scheduler(str(tmp_path), build.configs('mini2')[0].prediction_type)
    cfg.write_text(json.dumps(dict(sd_pndm, prediction_type='v_prediction')))
    with pytest.raises(NotImplementedError, match='v_prediction'):
        build.load_scheduler(str(tmp_path))
    cfg.write_text(json.dumps(dict(sd_pndm, _class_name='LMSDiscreteScheduler')))
    with pytest.raises(NotImplementedError, match='v_prediction'):
        build.load_scheduler(str(tmp_path), 'v_prediction')
    cfg.write_text(json.dumps({'_class_name': 'DDIMScheduler', 'beta_start': 0.00085, 'beta_end': 0.012, 'beta_schedule': 'scaled_linear',
                               'num_train_timesteps': 1000, 'clip_sample': False, 'set_alpha_to_one': False}))
    assert build.load_scheduler(str(tmp_path), 'v_prediction').config['prediction_type'] == 'v_prediction'
    # keys the file leaves out take DIFFUSERS' defaults, not this package's SD presets: a PNDM file without skip_prk_steps means the
    # Runge-Kutta warm-up (not provided: refused), one without betas means linear 1e-4 .. 0.02 (not provided by PNDM / LMS: refused)
    cfg.write_text(json.dumps({k: v for k, v in sd_pndm.items() if k != 'skip_prk_steps'}))
    with pytest.raises(NotImplementedError):
        build.load_scheduler(str(tmp_path))
    cfg.write_text(json.dumps({'_class_name': 'LMSDiscreteScheduler'}))
    with pytest.raises(NotImplementedError):
        build.load_scheduler(str(tmp_path))
    cfg.write_text(json.dumps({'_class_name': 'DDIMScheduler', 'clip_sample': False}))
    s = build.load_scheduler(str(tmp_path))
    assert s.config['beta_schedule'] == 'linear' and s.config['beta_end'] == 0.02 and s.config['set_alpha_to_one'] is True


def test_devmon_and_bench_clock_fields_without_a_gpu():
    '''tools/devmon.py never touches HIP and degrades to "no source" where no AMD GPU is visible; bench.devmon_collect
    turns whatever the sampler child printed into the `device.*` fields (None when there is nothing).'''
    import json
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'devmon.py'), '--probe'], capture_output=True, timeout=120)
    rec = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert set(rec) == {'source', 'errors', 'sample'}
    sys.path.insert(0, root)
    import bench
    assert bench.devmon_collect(None, 0.0, 1.0)['avg_sclk_mhz'] is None
    t0 = time.time()
    fake = subprocess.Popen([sys.executable, '-c',
                             'import sys, json, time; sys.stdin.read(); t = %r; '
                             'print(json.dumps({"source": "fake", "errors": [], "bdf": "0000:C5:00.0", "samples": '
                             '[[t + 0.1, 2000.0, 1000.0, 1900.0], [t + 0.2, 2100.0, 1100.0, 1900.0], [t + 9.0, 50.0, 90.0, 900.0]]}))' % t0],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    out = bench.devmon_collect(fake, t0, t0 + 1.0, '0000:c5:00.0')
    assert out['clock_source'] == 'fake' and out['clock_samples'] == 2
    assert out['avg_sclk_mhz'] == 2050.0 and out['avg_power_w'] == 1050.0 and out['max_sclk_mhz'] == 2100.0
    # the sampler says WHICH physical GPU it read, and the line says whether that is the timed one
    assert out['clock_device_bdf'] == '0000:C5:00.0' and out['clock_device_matches'] is True
    assert bench._bdf_key('c5:00.0') == bench._bdf_key('0000:C5:00.0') != bench._bdf_key('0000:c6:00.0')
    # amdsmi / sysfs ordinals are physical: the runtime's ordinal goes through ROCR_ then HIP_VISIBLE_DEVICES
    sys.path.insert(0, os.path.join(root, 'tools'))
    import devmon
    keep = {k: os.environ.pop(k, None) for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')}
    try:
        assert devmon.physical_index(1, 8) == 1
        os.environ['ROCR_VISIBLE_DEVICES'] = '4,5,6,7'
        assert devmon.physical_index(1, 8) == 5
        os.environ['HIP_VISIBLE_DEVICES'] = '2,3'
        assert devmon.physical_index(0, 8) == 6 and devmon.physical_index(1, 8) == 7
        os.environ['HIP_VISIBLE_DEVICES'] = 'GPU-deadbeef'
        assert devmon.physical_index(0, 8) == 0          # cannot be mapped here: the BDF comparison is the check
    finally:
        for k, v in keep.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_gn_fold_predicate_knows_the_kernel_limits():
    '''fd_groupnorm_fold_linear_f16 keeps a 16 x C fp16 tile in <= 48 KiB of LDS and 64 group slots; the host predicate must
    send wider layers to the unfused GroupNorm instead of letting the forward hit FD_ESHAPE.'''
    from flexdiffuse_amd import ops
    assert ops.gn_fold_supported(16, 4096, 320) and ops.gn_fold_supported(16, 4096, 320, 320, 32)
    assert ops.gn_fold_supported(2, 16384, 1536, 1536, 32)
    assert not ops.gn_fold_supported(2, 65536, 2048, 2048, 32)       # C too wide for the weight tile
    assert not ops.gn_fold_supported(16, 4096, 320, 320, 128)        # more groups than slots
    assert not ops.gn_fold_supported(1, 4096, 320)                   # B == 1: nothing shared
