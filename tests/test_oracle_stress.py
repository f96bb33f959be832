'''Which corners of `build.stress_unet_state_dict` can be parity targets at all -- decided by the fp32 ORACLE alone (CPU): the relative
change of the full-size SD1.5 UNet's output (32x32 latents) under a 2^-11 relative perturbation of its input, divided by 2^-11.
An fp16 implementation rounds every stored activation by up to 2^-11, so an amplification of ~1000 means the fp32 function itself
moves by ~100 % under fp16-sized noise: such a corner measures the chaos of random weights with one-hot softmax rows, not a kernel.'''
import torch


def _amplification(**kw) -> float:
    from flexdiffuse_amd import build
    from oracle import unet_ref
    ucfg = build.configs('sd15')[0]
    sd = {k: v.half().float() for k, v in build.stress_unet_state_dict('sd15', seed=3, **kw).items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn((1, 4, 32, 32), generator=g)
    ctx = torch.randn((1, 77, 768), generator=g).half().float()
    sign = torch.randn(x.shape, generator=g).sign()
    a = unet_ref.unet_forward(sd, ucfg, x, 500, ctx)
    b = unet_ref.unet_forward(sd, ucfg, x + 2 ** -11 * x.abs() * sign, 500, ctx)
    return float((a - b).abs().max() / a.abs().max()) / 2 ** -11


def test_stress_presets_are_well_conditioned_and_the_full_combination_is_not():
    amp_a = _amplification(branch_gain=1.0, qk_gain=1.5, gn_shift=6.0)      # preset A of tests/test_gpu_models.py
    amp_b = _amplification(branch_gain=0.25, qk_gain=4.0, gn_shift=6.0)     # preset B
    amp_all = _amplification(branch_gain=1.0, qk_gain=4.0, gn_shift=6.0)    # what VERDICT r4 proposed literally
    print(f'oracle amplification of a 2^-11 input perturbation: preset A {amp_a:.1f}, preset B {amp_b:.1f}, unit gain + q/k x 4: {amp_all:.0f}')
    assert amp_a < 10 and amp_b < 10
    assert amp_all > 100, 'the full combination became well-conditioned: make it the GPU parity preset'
