'''Assembly of the model containers (the part of the reference's `utils.Runner.__init__`,
utils.py:54-76, that is not model download): state dicts -> device containers -> pipeline.'''
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import weights as W
from .clip import CLIPModel
from .scheduler import DDIMScheduler
from .tokenizer import SyntheticTokenizer
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL

import dataclasses

# SD2.x-style small model: linear proj_in/out, head dim 64, v-prediction (BASELINE config 5 shape)
MINI2_UNET = dataclasses.replace(W.MINI_UNET, num_heads=(5, 10, 20), use_linear_projection=True,
                                 prediction_type='v_prediction')

PRESETS = {
    'sd15': (W.SD15_UNET, W.SD_VAE, W.CLIP_VIT_L14),
    'mini2': (MINI2_UNET, W.MINI_VAE, W.MINI_CLIP),
    'sd21': (W.SD21_UNET, W.SD_VAE, W.CLIP_VIT_H14),
    'mini': (W.MINI_UNET, W.MINI_VAE, W.MINI_CLIP),
}


def mini_unet_config(clip_cfg: W.CLIPConfig = W.MINI_CLIP) -> W.UNetConfig:
    import dataclasses
    return dataclasses.replace(W.MINI_UNET, cross_attention_dim=clip_cfg.text.hidden_size)


def synthetic_state_dicts(preset: str = 'sd15', seed: int = 0, branch_gain: float = 0.25,
                          parts=('unet', 'vae', 'clip')) -> Dict[str, dict]:
    '''Seeded fp32 CPU state dicts (HF key names) of the named architecture.'''
    ucfg, vcfg, ccfg = PRESETS[preset]
    if preset.startswith('mini'):
        ucfg = dataclasses.replace(ucfg, cross_attention_dim=ccfg.text.hidden_size)
    out = {}
    if 'unet' in parts:
        out['unet'] = W.synth_state_dict(W.unet_param_shapes(ucfg), seed, branch_gain, 'unet.')
    if 'vae' in parts:
        out['vae'] = W.synth_state_dict(W.vae_param_shapes(vcfg), seed, branch_gain, 'vae.')
    if 'clip' in parts:
        out['clip'] = W.synth_state_dict(W.clip_param_shapes(ccfg), seed, 1.0, 'clip.')
    return out


def configs(preset: str):
    ucfg, vcfg, ccfg = PRESETS[preset]
    if preset.startswith('mini'):
        ucfg = dataclasses.replace(ucfg, cross_attention_dim=ccfg.text.hidden_size)
    return ucfg, vcfg, ccfg


def build_models(state_dicts: Dict[str, dict], preset: str = 'sd15', device='cuda',
                 vae_encoder: bool = True, steps_offset: int = 0):
    '''(pipeline, clip, tokenizer): device containers + FlexPipeline around them.'''
    from .pipeline.flex import FlexPipeline
    ucfg, vcfg, ccfg = configs(preset)
    unet = UNet2DConditionModel(state_dicts['unet'], ucfg, device)
    vae = AutoencoderKL(state_dicts['vae'], vcfg, device, encoder=vae_encoder)
    clip = CLIPModel(state_dicts['clip'], ccfg, device)
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size,
                             model_max_length=ccfg.text.max_position_embeddings)
    sched = DDIMScheduler(steps_offset=steps_offset, prediction_type=ucfg.prediction_type)
    pipe = FlexPipeline(vae, clip, tok, unet, sched).to(device)
    return pipe, clip, tok
