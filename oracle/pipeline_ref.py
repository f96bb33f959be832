'''CPU oracle: the denoising loop of the reference in torch fp32.

TEST INFRASTRUCTURE ONLY.  Follows the reference line by line where the reference owns the
arithmetic -- pipeline/flex.py:170-310 (timesteps, init latents / img2img noise level,
t_start, loop, decode scaling 1/0.18215, (x/2+0.5).clamp(0,1)) and pipeline/guide.py:46-64
(classifier-free guidance: [uncond]*B + embeds, duplicated latents, u + g (t - u)) -- and
calls the UNPINNED UNet / VAE / DDIM restatements for the third-party parts.
'''
import torch

from . import ddim_ref, unet_ref, vae_ref

VAE_SCALE = 0.18215


@torch.no_grad()
def noise_pred(sd_unet, ucfg, latents, t, embeds, uncond, guidance, unet_fn=None):
    '''pipeline/guide.py:46-64.  PINNED (this function, not the UNet it calls) by
    tests/golden/backhalf_goldens.npz `simple/*`: the reference's own SimpleGuide.noise_pred run
    with a recording stub UNet (tests/golden/make_backhalf_goldens.py); `unet_fn(latents, t, ctx)`
    replays that stub in tests/test_oracle_backhalf.py.'''
    if unet_fn is None:
        unet_fn = lambda x, tt, ctx: unet_ref.unet_forward(sd_unet, ucfg, x, tt, ctx)
    B = latents.shape[0]
    if guidance > 1.0:
        ctx = torch.cat([uncond.expand(B, -1, -1), embeds])
        out = unet_fn(torch.cat([latents] * 2), t, ctx)
        u, c = out.chunk(2)
        return u + guidance * (c - u)
    return unet_fn(latents, t, embeds)


@torch.no_grad()
def denoise(sd_unet, ucfg, embeds, uncond, latents, steps, guidance, steps_offset=0, t_start=0,
            callback=None, eta=0.0, noise_fn=None):
    '''pipeline/flex.py:262-287 with DDIM.  Returns (final latents, timesteps used).  eta > 0
    (pipeline/flex.py:247-251): `noise_fn(shape)` supplies the per-step variance noise.'''
    acp = ddim_ref.alphas_cumprod()
    ts = ddim_ref.timesteps(steps, steps_offset=steps_offset)
    used = []
    x = latents.float().clone()
    for t in ts[t_start:]:
        eps = noise_pred(sd_unet, ucfg, x, int(t), embeds.float(), uncond.float(), guidance)
        x = ddim_ref.ddim_step(eps, int(t), x, acp, steps,
                               prediction_type=getattr(ucfg, 'prediction_type', 'epsilon'), eta=eta,
                               noise=noise_fn(tuple(x.shape)) if eta else None)
        used.append(int(t))
        if callback:
            callback(int(t), x)
    return x, used


@torch.no_grad()
def img2img_init(sd_vae, vcfg, image, posterior_noise, noise, steps, strength, batch_size,
                 steps_offset=0):
    '''pipeline/flex.py:181-221: encode, sample, scale, repeat, add noise; returns
    (noisy latents, t_start).'''
    mean, logvar = vae_ref.vae_encode_moments(sd_vae, vcfg, image)
    z = vae_ref.vae_sample(mean, logvar, posterior_noise) * VAE_SCALE
    z = torch.cat([z] * batch_size)
    init_timestep = min(int(steps * strength) + steps_offset, steps)
    ts = ddim_ref.timesteps(steps, steps_offset=steps_offset)
    t = int(ts[-init_timestep])
    z = ddim_ref.add_noise(z, noise, t, ddim_ref.alphas_cumprod())
    return z, max(steps - init_timestep + steps_offset, 0)


@torch.no_grad()
def decode_image(sd_vae, vcfg, latents):
    '''pipeline/flex.py:112-124 up to the clamp: (B,3,H,W) in [0,1].'''
    img = vae_ref.vae_decode(sd_vae, vcfg, latents.float() / VAE_SCALE)
    return (img / 2 + 0.5).clamp(0, 1)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    '''min over the batch of 10 log10(1 / MSE) for images in [0,1].'''
    mse = ((a.float() - b.float()) ** 2).flatten(1).mean(dim=1).clamp_min(1e-20)
    return float((10.0 * torch.log10(1.0 / mse)).min())
