'''CPU oracle: Stable-Diffusion VAE (AutoencoderKL) encode / decode in torch fp32.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference calls diffusers==0.3.0
`AutoencoderKL` (call sites pipeline/flex.py:118,189-191); restated from the published
architecture, anchored by the exact parameter count 83,653,863.
'''
import torch
import torch.nn.functional as F

from .unet_ref import _conv, _gn, _lin, resnet


def _attn_block(sd, name, x, groups):
    '''diffusers 0.3.0 AttentionBlock, single head: softmax((q s)(k s)^T) v, s=C^-1/4.'''
    B, C, H, W = x.shape
    h = _gn(sd, name + '.group_norm', x, groups, 1e-6).reshape(B, C, H * W).transpose(1, 2)
    q, k, v = (_lin(sd, f'{name}.{n}', h) for n in ('query', 'key', 'value'))
    scale = 1.0 / (C ** 0.25)
    p = ((q * scale) @ (k * scale).transpose(-1, -2)).softmax(dim=-1)
    o = _lin(sd, name + '.proj_attn', p @ v)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


@torch.no_grad()
def vae_decode(sd, cfg, z: torch.Tensor) -> torch.Tensor:
    '''latents (B,4,h,w) -> image (B,3,8h,8w); caller applies 1/0.18215 first.'''
    G = cfg.norm_num_groups
    rev = list(reversed(cfg.block_out_channels))
    h = _conv(sd, 'post_quant_conv', z.float(), padding=0)
    h = _conv(sd, 'decoder.conv_in', h)
    h = resnet(sd, 'decoder.mid_block.resnets.0', h, None, G, 1e-6)
    h = _attn_block(sd, 'decoder.mid_block.attentions.0', h, G)
    h = resnet(sd, 'decoder.mid_block.resnets.1', h, None, G, 1e-6)
    for i in range(len(rev)):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(sd, f'decoder.up_blocks.{i}.resnets.{j}', h, None, G, 1e-6)
        if i != len(rev) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = _conv(sd, f'decoder.up_blocks.{i}.upsamplers.0.conv', h)
    h = F.silu(_gn(sd, 'decoder.conv_norm_out', h, G, 1e-6))
    return _conv(sd, 'decoder.conv_out', h)


@torch.no_grad()
def vae_encode_moments(sd, cfg, x: torch.Tensor):
    '''image (B,3,H,W) in [-1,1] -> (mean, logvar) of the latent distribution.'''
    G = cfg.norm_num_groups
    ch = cfg.block_out_channels
    h = _conv(sd, 'encoder.conv_in', x.float())
    for i in range(len(ch)):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, f'encoder.down_blocks.{i}.resnets.{j}', h, None, G, 1e-6)
        if i != len(ch) - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = _conv(sd, f'encoder.down_blocks.{i}.downsamplers.0.conv', h, stride=2,
                      padding=0)
    h = resnet(sd, 'encoder.mid_block.resnets.0', h, None, G, 1e-6)
    h = _attn_block(sd, 'encoder.mid_block.attentions.0', h, G)
    h = resnet(sd, 'encoder.mid_block.resnets.1', h, None, G, 1e-6)
    h = F.silu(_gn(sd, 'encoder.conv_norm_out', h, G, 1e-6))
    h = _conv(sd, 'encoder.conv_out', h)
    mean, logvar = _conv(sd, 'quant_conv', h, padding=0).chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_sample(mean, logvar, noise):
    '''DiagonalGaussianDistribution.sample: mean + exp(0.5 logvar) * noise.'''
    return mean + torch.exp(0.5 * logvar) * noise
