'''Known-answer tests for the architecture restatement: exact published parameter
counts and deterministic synthetic weights.  CPU only.'''
import torch

from flexdiffuse_amd import weights as W


def test_parameter_counts_exact():
    assert W.count_params(W.unet_param_shapes(W.SD15_UNET)) == 859_520_964
    assert W.count_params(W.unet_param_shapes(W.SD21_UNET)) == 865_910_724
    vae = W.vae_param_shapes(W.SD_VAE)
    assert W.count_params(vae) == 83_653_863
    enc = {k: v for k, v in vae.items() if k.startswith('encoder.')}
    dec = {k: v for k, v in vae.items() if k.startswith('decoder.')}
    assert W.count_params(enc) == 34_163_592 and W.count_params(dec) == 49_490_179
    clip = W.clip_param_shapes(W.CLIP_VIT_L14)
    assert W.count_params(clip) == 427_616_513
    assert W.count_params({k: v for k, v in clip.items() if k.startswith('text_model')}) \
        == 123_060_480


def test_up_plan_channels():
    plan = W.unet_up_plan(W.SD15_UNET)
    cins = [[a + b for a, b in res] for res, _, _, _ in plan]
    assert cins == [[2560, 2560, 2560], [2560, 2560, 1920], [1920, 1280, 960], [960, 640, 640]]


def test_synth_weights_deterministic():
    shapes = W.clip_param_shapes(W.MINI_CLIP)
    a = W.synth_state_dict(shapes, seed=3)
    b = W.synth_state_dict(shapes, seed=3)
    c = W.synth_state_dict(shapes, seed=4)
    k = 'text_model.encoder.layers.0.mlp.fc1.weight'
    assert torch.equal(a[k], b[k]) and not torch.equal(a[k], c[k])
    assert a['text_model.encoder.layers.0.layer_norm1.weight'].mean().item() > 0.8


def test_load_state_dicts_from_safetensors(tmp_path):
    '''Checkpoint loading: a diffusers-layout directory written with
    safetensors round-trips through build.load_state_dicts, including the newer VAE attention
    key names and 1x1-conv attention weights; a wrong architecture fails loudly.'''
    import os
    import pytest
    import torch
    from safetensors.torch import save_file
    from flexdiffuse_amd import build
    sds = build.synthetic_state_dicts('mini', seed=3)
    sd_dir, clip_dir = tmp_path / 'sd', tmp_path / 'clip'
    for sub in ('unet', 'vae'):
        os.makedirs(sd_dir / sub)
    os.makedirs(clip_dir)
    save_file({k: v.contiguous() for k, v in sds['unet'].items()},
              str(sd_dir / 'unet' / 'diffusion_pytorch_model.safetensors'))
    vae = {}
    for k, v in sds['vae'].items():          # write the VAE in the newer naming / conv form
        for old, new in (('.query.', '.to_q.'), ('.key.', '.to_k.'), ('.value.', '.to_v.'),
                         ('.proj_attn.', '.to_out.0.')):
            if old in k:
                k = k.replace(old, new)
                if k.endswith('.weight'):
                    v = v.reshape(*v.shape, 1, 1)
        vae[k] = v.contiguous()
    save_file(vae, str(sd_dir / 'vae' / 'diffusion_pytorch_model.safetensors'))
    clip = {k: v.contiguous() for k, v in sds['clip'].items()}
    clip['text_model.embeddings.position_ids'] = torch.arange(8).unsqueeze(0)
    save_file(clip, str(clip_dir / 'model.safetensors'))
    got = build.load_state_dicts(str(sd_dir), str(clip_dir), 'mini')
    for part in ('unet', 'vae', 'clip'):
        assert list(got[part]) == list(sds[part])
        for k in sds[part]:
            assert torch.equal(got[part][k], sds[part][k]), (part, k)
    with pytest.raises(ValueError, match='does not match the architecture'):
        build.load_state_dicts(str(sd_dir), str(clip_dir), 'sd15')


def test_load_state_dicts_from_bin(tmp_path):
    '''The torch-pickle layout of the 2022 repositories (diffusion_pytorch_model.bin,
    pytorch_model.bin) loads the same way; a pickle that is not a flat tensor dict is refused.'''
    import os
    import pytest
    import torch
    from flexdiffuse_amd import build
    sds = build.synthetic_state_dicts('mini', seed=4)
    sd_dir, clip_dir = tmp_path / 'sd', tmp_path / 'clip'
    for sub in ('unet', 'vae'):
        os.makedirs(sd_dir / sub)
        torch.save({k: v.half() for k, v in sds[sub].items()}, str(sd_dir / sub / 'diffusion_pytorch_model.bin'))
    os.makedirs(clip_dir)
    torch.save(dict(sds['clip']), str(clip_dir / 'pytorch_model.bin'))
    got = build.load_state_dicts(str(sd_dir), str(clip_dir), 'mini')
    for k, v in sds['unet'].items():
        assert got['unet'][k].dtype == torch.float32 and torch.equal(got['unet'][k], v.half().float()), k
    for k, v in sds['clip'].items():
        assert torch.equal(got['clip'][k], v), k
    torch.save({'state_dict': dict(sds['clip'])}, str(clip_dir / 'pytorch_model.bin'))
    with pytest.raises(ValueError, match='flat name'):
        build.load_state_dicts(str(sd_dir), str(clip_dir), 'mini')
