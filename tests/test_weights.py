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
