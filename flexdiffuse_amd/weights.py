'''Architecture configs, parameter inventories (Hugging Face key names) and seeded
synthetic state-dicts for the Stable-Diffusion UNet, VAE and CLIP towers.

Neither box has model weights or network access, so benchmarks and
parity tests run on seeded synthetic weights of the exact architecture.  The parameter
enumerations below reproduce the published counts exactly (tests/test_weights.py):
SD-v1 UNet 859,520,964; SD2 UNet 865,910,724; VAE 83,653,863; CLIP ViT-L/14
427,616,513 (text tower 123,060,480).  A real checkpoint with HF key names loads
through the same `load_state_dict` paths of the model containers.
'''
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------- configs
@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    cross_attn: Tuple[bool, ...] = (True, True, True, False)   # per down block
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)                  # SD1.x "attention_head_dim=8"
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    use_linear_projection: bool = False                        # SD2.x: linear proj_in/out
    prediction_type: str = 'epsilon'

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


SD15_UNET = UNetConfig()
SD21_UNET = UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                       use_linear_projection=True, prediction_type='v_prediction')
# small UNet exercising every kernel template (head dims 40 / 80 / 160, concat skips,
# up/down-sampling, shortcuts) at a size the CPU oracle finishes in seconds
MINI_UNET = UNetConfig(block_out_channels=(320, 640, 1280), cross_attn=(True, True, False),
                       layers_per_block=1, num_heads=(8, 8, 8))


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215   # pipeline/flex.py:117,192


SD_VAE = VAEConfig()
MINI_VAE = VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = 'quick_gelu'


@dataclass(frozen=True)
class CLIPVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    hidden_act: str = 'quick_gelu'

    @property
    def num_positions(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


@dataclass(frozen=True)
class CLIPConfig:
    text: CLIPTextConfig = field(default_factory=CLIPTextConfig)
    vision: CLIPVisionConfig = field(default_factory=CLIPVisionConfig)
    projection_dim: int = 768


CLIP_VIT_L14 = CLIPConfig()
# OpenCLIP ViT-H/14 dims (SD2.1, BASELINE config 5)
CLIP_VIT_H14 = CLIPConfig(
    text=CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                        num_attention_heads=16, hidden_act='gelu'),
    vision=CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32,
                            num_attention_heads=16, hidden_act='gelu'),
    projection_dim=1024)
MINI_CLIP = CLIPConfig(
    text=CLIPTextConfig(vocab_size=512, hidden_size=128, intermediate_size=256,
                        num_hidden_layers=2, num_attention_heads=2),
    vision=CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=2),
    projection_dim=128)   # must equal the text width (guide tokens are blended with text)


# ------------------------------------------------------------------- parameter shapes
Shapes = 'OrderedDict[str, Tuple[int, ...]]'


def _lin(d: dict, name: str, cin: int, cout: int, bias: bool = True):
    d[name + '.weight'] = (cout, cin)
    if bias:
        d[name + '.bias'] = (cout,)


def _conv(d: dict, name: str, cin: int, cout: int, k: int, bias: bool = True):
    d[name + '.weight'] = (cout, cin, k, k)
    if bias:
        d[name + '.bias'] = (cout,)


def _norm(d: dict, name: str, c: int):
    d[name + '.weight'] = (c,)
    d[name + '.bias'] = (c,)


def _resnet(d: dict, name: str, cin: int, cout: int, temb: Optional[int]):
    _norm(d, name + '.norm1', cin)
    _conv(d, name + '.conv1', cin, cout, 3)
    if temb:
        _lin(d, name + '.time_emb_proj', temb, cout)
    _norm(d, name + '.norm2', cout)
    _conv(d, name + '.conv2', cout, cout, 3)
    if cin != cout:
        _conv(d, name + '.conv_shortcut', cin, cout, 1)


def _transformer(d: dict, name: str, c: int, ctx: int, linear_proj: bool):
    _norm(d, name + '.norm', c)
    if linear_proj:
        _lin(d, name + '.proj_in', c, c)
    else:
        _conv(d, name + '.proj_in', c, c, 1)
    tb = name + '.transformer_blocks.0'
    for ln in ('norm1', 'norm2', 'norm3'):
        _norm(d, f'{tb}.{ln}', c)
    for attn, kv in (('attn1', c), ('attn2', ctx)):
        _lin(d, f'{tb}.{attn}.to_q', c, c, bias=False)
        _lin(d, f'{tb}.{attn}.to_k', kv, c, bias=False)
        _lin(d, f'{tb}.{attn}.to_v', kv, c, bias=False)
        _lin(d, f'{tb}.{attn}.to_out.0', c, c)
    _lin(d, f'{tb}.ff.net.0.proj', c, 8 * c)
    _lin(d, f'{tb}.ff.net.2', 4 * c, c)
    if linear_proj:
        _lin(d, name + '.proj_out', c, c)
    else:
        _conv(d, name + '.proj_out', c, c, 1)


def unet_param_shapes(cfg: UNetConfig = SD15_UNET) -> 'OrderedDict[str, tuple]':
    '''diffusers `UNet2DConditionModel` parameters.'''
    d: 'OrderedDict[str, tuple]' = OrderedDict()
    ch = cfg.block_out_channels
    temb = cfg.time_embed_dim
    _conv(d, 'conv_in', cfg.in_channels, ch[0], 3)
    _lin(d, 'time_embedding.linear_1', ch[0], temb)
    _lin(d, 'time_embedding.linear_2', temb, temb)
    cur = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(d, f'down_blocks.{i}.resnets.{j}', cur, c, temb)
            cur = c
            if cfg.cross_attn[i]:
                _transformer(d, f'down_blocks.{i}.attentions.{j}', c, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        if i != len(ch) - 1:
            _conv(d, f'down_blocks.{i}.downsamplers.0.conv', c, c, 3)
    _resnet(d, 'mid_block.resnets.0', cur, cur, temb)
    _transformer(d, 'mid_block.attentions.0', cur, cfg.cross_attention_dim,
                 cfg.use_linear_projection)
    _resnet(d, 'mid_block.resnets.1', cur, cur, temb)
    for i, (skips, c, has_attn, has_up) in enumerate(unet_up_plan(cfg)):
        for j, (cin_x, cin_skip) in enumerate(skips):
            _resnet(d, f'up_blocks.{i}.resnets.{j}', cin_x + cin_skip, c, temb)
            if has_attn:
                _transformer(d, f'up_blocks.{i}.attentions.{j}', c, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        if has_up:
            _conv(d, f'up_blocks.{i}.upsamplers.0.conv', c, c, 3)
    _norm(d, 'conv_norm_out', ch[0])
    _conv(d, 'conv_out', ch[0], cfg.out_channels, 3)
    return d


def unet_skip_channels(cfg: UNetConfig):
    '''Channel count of every residual the down path pushes (conv_in first).'''
    ch = cfg.block_out_channels
    skips = [ch[0]]
    for i, c in enumerate(ch):
        skips += [c] * cfg.layers_per_block
        if i != len(ch) - 1:
            skips.append(c)
    return skips


def unet_up_plan(cfg: UNetConfig):
    '''Per up block: ([(x_channels, skip_channels) per resnet], out_channels, has_attn,
    has_upsampler) -- the skip stack is popped from the end.'''
    ch = cfg.block_out_channels
    skips = unet_skip_channels(cfg)
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.cross_attn))
    plan = []
    cur = ch[-1]
    for i, c in enumerate(rev):
        res = []
        for _ in range(cfg.layers_per_block + 1):
            res.append((cur, skips.pop()))
            cur = c
        plan.append((res, c, rev_attn[i], i != len(rev) - 1))
    assert not skips
    return plan


def unet_heads_for_channels(cfg: UNetConfig) -> Dict[int, int]:
    return {c: h for c, h in zip(cfg.block_out_channels, cfg.num_heads)}


def _vae_attn(d: dict, name: str, c: int):
    _norm(d, name + '.group_norm', c)
    for n in ('query', 'key', 'value', 'proj_attn'):
        _lin(d, f'{name}.{n}', c, c)


def vae_param_shapes(cfg: VAEConfig = SD_VAE) -> 'OrderedDict[str, tuple]':
    '''diffusers `AutoencoderKL` parameters.'''
    d: 'OrderedDict[str, tuple]' = OrderedDict()
    ch = cfg.block_out_channels
    # encoder
    _conv(d, 'encoder.conv_in', cfg.in_channels, ch[0], 3)
    cur = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(d, f'encoder.down_blocks.{i}.resnets.{j}', cur, c, None)
            cur = c
        if i != len(ch) - 1:
            _conv(d, f'encoder.down_blocks.{i}.downsamplers.0.conv', c, c, 3)
    _resnet(d, 'encoder.mid_block.resnets.0', cur, cur, None)
    _vae_attn(d, 'encoder.mid_block.attentions.0', cur)
    _resnet(d, 'encoder.mid_block.resnets.1', cur, cur, None)
    _norm(d, 'encoder.conv_norm_out', cur)
    _conv(d, 'encoder.conv_out', cur, 2 * cfg.latent_channels, 3)
    # decoder
    rev = list(reversed(ch))
    _conv(d, 'decoder.conv_in', cfg.latent_channels, rev[0], 3)
    cur = rev[0]
    _resnet(d, 'decoder.mid_block.resnets.0', cur, cur, None)
    _vae_attn(d, 'decoder.mid_block.attentions.0', cur)
    _resnet(d, 'decoder.mid_block.resnets.1', cur, cur, None)
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet(d, f'decoder.up_blocks.{i}.resnets.{j}', cur, c, None)
            cur = c
        if i != len(rev) - 1:
            _conv(d, f'decoder.up_blocks.{i}.upsamplers.0.conv', c, c, 3)
    _norm(d, 'decoder.conv_norm_out', cur)
    _conv(d, 'decoder.conv_out', cur, cfg.out_channels, 3)
    _conv(d, 'quant_conv', 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    _conv(d, 'post_quant_conv', cfg.latent_channels, cfg.latent_channels, 1)
    return d


def _clip_layers(d: dict, prefix: str, n: int, w: int, mlp: int):
    for i in range(n):
        p = f'{prefix}.encoder.layers.{i}'
        for proj in ('k_proj', 'v_proj', 'q_proj', 'out_proj'):
            _lin(d, f'{p}.self_attn.{proj}', w, w)
        _norm(d, f'{p}.layer_norm1', w)
        _lin(d, f'{p}.mlp.fc1', w, mlp)
        _lin(d, f'{p}.mlp.fc2', mlp, w)
        _norm(d, f'{p}.layer_norm2', w)


def clip_param_shapes(cfg: CLIPConfig = CLIP_VIT_L14) -> 'OrderedDict[str, tuple]':
    '''transformers `CLIPModel` parameters.'''
    d: 'OrderedDict[str, tuple]' = OrderedDict()
    t, v = cfg.text, cfg.vision
    d['logit_scale'] = ()
    d['text_model.embeddings.token_embedding.weight'] = (t.vocab_size, t.hidden_size)
    d['text_model.embeddings.position_embedding.weight'] = (t.max_position_embeddings,
                                                            t.hidden_size)
    _clip_layers(d, 'text_model', t.num_hidden_layers, t.hidden_size, t.intermediate_size)
    _norm(d, 'text_model.final_layer_norm', t.hidden_size)
    d['vision_model.embeddings.class_embedding'] = (v.hidden_size,)
    d['vision_model.embeddings.patch_embedding.weight'] = (v.hidden_size, 3, v.patch_size,
                                                           v.patch_size)
    d['vision_model.embeddings.position_embedding.weight'] = (v.num_positions, v.hidden_size)
    _norm(d, 'vision_model.pre_layrnorm', v.hidden_size)
    _clip_layers(d, 'vision_model', v.num_hidden_layers, v.hidden_size, v.intermediate_size)
    _norm(d, 'vision_model.post_layernorm', v.hidden_size)
    d['visual_projection.weight'] = (cfg.projection_dim, v.hidden_size)
    d['text_projection.weight'] = (cfg.projection_dim, t.hidden_size)
    return d


def count_params(shapes) -> int:
    n = 0
    for s in shapes.values():
        k = 1
        for v in s:
            k *= v
        n += k
    return n


# ------------------------------------------------------------------ synthetic weights
# Residual-branch output layers get a reduced gain so that a 50-step, guidance-scale-8
# denoising loop on random weights stays well-conditioned (fp16 vs fp32 drift bounded),
# while every layer still contributes to the output (no zero-init: a zeroed branch
# would make parity checks vacuous).
_BRANCH_OUT = ('.conv2.weight', '.to_out.0.weight', '.ff.net.2.weight', '.proj_out.weight',
               '.proj_attn.weight', '.out_proj.weight', '.mlp.fc2.weight')


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0,
                 branch_gain: float = 0.25) -> torch.Tensor:
    gen = torch.Generator('cpu').manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1))
                                             & 0x7FFFFFFF)
    shape = tuple(shape)
    if name == 'logit_scale':
        return torch.tensor(4.6052)
    if name.endswith('class_embedding'):
        return torch.randn(shape, generator=gen) * 0.02
    if 'embedding.weight' in name and 'patch' not in name:
        return torch.randn(shape, generator=gen) * 0.02
    is_norm = ('norm' in name or 'layrnorm' in name) and len(shape) == 1
    if is_norm and name.endswith('.weight'):
        return 1.0 + 0.1 * torch.randn(shape, generator=gen)
    if name.endswith('.bias'):
        return 0.05 * torch.randn(shape, generator=gen)
    fan_in = 1
    for v in shape[1:]:
        fan_in *= v
    gain = 1.0
    if name.endswith(_BRANCH_OUT):
        gain = branch_gain
    if name.endswith('time_emb_proj.weight'):
        gain = 0.5
    return torch.randn(shape, generator=gen) * (gain / fan_in ** 0.5)


def synth_state_dict(shapes, seed: int = 0, branch_gain: float = 0.25,
                     prefix: str = '') -> 'OrderedDict[str, torch.Tensor]':
    '''Seeded fp32 CPU state-dict with the given shapes; deterministic per (name, seed).'''
    return OrderedDict((k, synth_tensor(prefix + k, s, seed, branch_gain))
                       for k, s in shapes.items())
