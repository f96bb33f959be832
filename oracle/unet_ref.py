'''CPU oracle: Stable-Diffusion UNet forward in torch fp32.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference calls
`diffusers==0.3.0 UNet2DConditionModel` (requirements.txt:1; call sites
pipeline/guide.py:56-58, composition/guide.py:62-64) which is neither vendored nor
installed here.  This restates the published architecture and is
anchored by the exact parameter count (859,520,964 for SD-v1) in tests/test_weights.py.
Takes a plain state-dict with Hugging Face key names and a config object exposing
block_out_channels / cross_attn / layers_per_block / num_heads / norm_num_groups /
use_linear_projection.
'''
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    '''diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0).'''
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride=stride,
                    padding=padding)


def _lin(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def _gn(sd, name, x, groups, eps):
    return F.group_norm(x, groups, sd[name + '.weight'], sd[name + '.bias'], eps)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], 1e-5)


def resnet(sd, name, x, temb, groups, eps=1e-5):
    h = _conv(sd, name + '.conv1', F.silu(_gn(sd, name + '.norm1', x, groups, eps)))
    if temb is not None and (name + '.time_emb_proj.weight') in sd:
        h = h + _lin(sd, name + '.time_emb_proj', F.silu(temb))[:, :, None, None]
    h = _conv(sd, name + '.conv2', F.silu(_gn(sd, name + '.norm2', h, groups, eps)))
    if (name + '.conv_shortcut.weight') in sd:
        x = _conv(sd, name + '.conv_shortcut', x, padding=0)
    return x + h


def attention(q, k, v, heads, mask=None):
    '''(B,Nq,C),(B,Nk,C),(B,Nk,C) -> (B,Nq,C); scale = head_dim**-0.5.'''
    B, Nq, C = q.shape
    d = C // heads
    q = q.view(B, Nq, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    if mask is not None:
        s = s + mask
    o = s.softmax(dim=-1) @ v
    return o.transpose(1, 2).reshape(B, Nq, C)


def _cross_attn(sd, name, x, ctx, heads):
    q = _lin(sd, name + '.to_q', x)
    k = _lin(sd, name + '.to_k', ctx)
    v = _lin(sd, name + '.to_v', ctx)
    return _lin(sd, name + '.to_out.0', attention(q, k, v, heads))


def transformer(sd, name, x, ctx, heads, groups, linear_proj=False):
    B, C, H, W = x.shape
    res = x
    h = _gn(sd, name + '.norm', x, groups, 1e-6)
    if linear_proj:
        h = _lin(sd, name + '.proj_in', h.permute(0, 2, 3, 1).reshape(B, H * W, C))
    else:
        h = _conv(sd, name + '.proj_in', h, padding=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    tb = name + '.transformer_blocks.0'
    h = _cross_attn(sd, tb + '.attn1', _ln(sd, tb + '.norm1', h), _ln(sd, tb + '.norm1', h),
                    heads) + h
    h = _cross_attn(sd, tb + '.attn2', _ln(sd, tb + '.norm2', h), ctx, heads) + h
    g = _lin(sd, tb + '.ff.net.0.proj', _ln(sd, tb + '.norm3', h))
    a, gate = g.chunk(2, dim=-1)
    h = _lin(sd, tb + '.ff.net.2', a * F.gelu(gate)) + h
    if linear_proj:
        h = _lin(sd, name + '.proj_out', h).reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = _conv(sd, name + '.proj_out', h.reshape(B, H, W, C).permute(0, 3, 1, 2), padding=0)
    return h + res


@torch.no_grad()
def unet_forward(sd, cfg, sample: torch.Tensor, timestep, ctx: torch.Tensor) -> torch.Tensor:
    '''(B,4,h,w) fp32, scalar/(B,) timestep, (B,77,D) -> (B,4,h,w).'''
    sample, ctx = sample.float(), ctx.float()
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    ch = cfg.block_out_channels
    G = cfg.norm_num_groups
    heads = dict(zip(ch, cfg.num_heads))
    emb = timestep_embedding(t, ch[0])
    emb = _lin(sd, 'time_embedding.linear_2', F.silu(_lin(sd, 'time_embedding.linear_1', emb)))
    h = _conv(sd, 'conv_in', sample)
    skips = [h]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, f'down_blocks.{i}.resnets.{j}', h, emb, G)
            if cfg.cross_attn[i]:
                h = transformer(sd, f'down_blocks.{i}.attentions.{j}', h, ctx, heads[c], G,
                                cfg.use_linear_projection)
            skips.append(h)
        if i != len(ch) - 1:
            h = _conv(sd, f'down_blocks.{i}.downsamplers.0.conv', h, stride=2)
            skips.append(h)
    mc = ch[-1]
    h = resnet(sd, 'mid_block.resnets.0', h, emb, G)
    h = transformer(sd, 'mid_block.attentions.0', h, ctx, heads[mc], G,
                    cfg.use_linear_projection)
    h = resnet(sd, 'mid_block.resnets.1', h, emb, G)
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.cross_attn))
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, f'up_blocks.{i}.resnets.{j}', h, emb, G)
            if rev_attn[i]:
                h = transformer(sd, f'up_blocks.{i}.attentions.{j}', h, ctx, heads[c], G,
                                cfg.use_linear_projection)
        if i != len(rev) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = _conv(sd, f'up_blocks.{i}.upsamplers.0.conv', h)
    h = F.silu(_gn(sd, 'conv_norm_out', h, G, 1e-5))
    return _conv(sd, 'conv_out', h)
