'''CLIP text + vision towers on gfx950 -- stands where the reference passes transformers'
`CLIPModel` and exposes exactly the surface encode/clip.py:64-100 touches:
`clip.text_model(ids)[0]`, `clip.vision_model.{embeddings, pre_layrnorm, encoder,
post_layernorm}`, `clip.visual_projection`, `clip.device`.

Both towers run on the shared HIP kernels: LayerNorm, MFMA GEMM with fused bias /
quick-GELU / residual epilogues, and the flash attention kernel (head dim 64; causal for
text).  Hidden states are fp16 on device; the final outputs (text last_hidden_state,
projected image tokens) are returned as fp32 like the reference's tensors.
'''
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch

from . import hip, ops
from .weights import CLIP_VIT_L14, CLIPConfig, clip_param_shapes


class _Layer:
    def __init__(self, sd, p, dev):
        lin = lambda n: ops.prep_linear(sd[f'{p}.{n}.weight'], sd[f'{p}.{n}.bias'], dev)
        ln = lambda n: (ops.f32(sd[f'{p}.{n}.weight'], dev), ops.f32(sd[f'{p}.{n}.bias'], dev))
        self.ln1, self.ln2 = ln('layer_norm1'), ln('layer_norm2')
        self.q, self.k, self.v = lin('self_attn.q_proj'), lin('self_attn.k_proj'), lin('self_attn.v_proj')
        self.o = lin('self_attn.out_proj')
        self.fc1, self.fc2 = lin('mlp.fc1'), lin('mlp.fc2')


def _encoder_forward(layers, h: torch.Tensor, B: int, T: int, heads: int, act: int,
                     causal: bool) -> torch.Tensor:
    '''h [B*T][C] fp16 through pre-LN transformer layers.'''
    C = h.shape[1]
    d = C // heads
    ldv = (T + 7) // 8 * 8
    for l in layers:
        n = ops.layernorm(h, *l.ln1)
        q, k = ops.gemm(n, l.q), ops.gemm(n, l.k)
        vt = ops.gemm_vt(n, l.v, B, T, ldv)
        o = ops.attention(q, k, vt, B, heads, T, T, d, causal)
        h = ops.gemm(o, l.o, residual=h)
        n = ops.layernorm(h, *l.ln2)
        h = ops.gemm(ops.gemm(n, l.fc1, act=act), l.fc2, residual=h)
    return h


class _PlannedEncoder:
    '''The transformer stack of a tower behind a launch plan (hip.Plan), one per (batch, tokens): ~10 launches per layer x 12 / 24 layers
    issued by ONE library call instead of ~360 trips through the Python front -- the towers' kernels take 3-4 ms per request, the
    eager front 5-15 ms of host time depending on the box (profiles/r06_session_ab.txt sec. 8), during which the device idles.  Same
    kernels, same order, same bits (tests/test_gpu_models.py::test_clip_towers_through_their_launch_plans).  FD_CLIP_PLAN=0: eager.'''
    MAX_PLANS = 4

    def __init__(self, layers, heads: int, act: int, causal: bool):
        self.layers, self.heads, self.act, self.causal = layers, heads, act, causal
        self._plans = {}

    def __call__(self, h: torch.Tensor, B: int, T: int) -> torch.Tensor:
        import os
        if os.environ.get('FD_CLIP_PLAN', '1') == '0' or torch.cuda.is_current_stream_capturing():
            return _encoder_forward(self.layers, h, B, T, self.heads, self.act, self.causal)
        key = (B, T, h.device.index, hip.stream().value)
        e = self._plans.get(key)
        if e is None:
            _encoder_forward(self.layers, h, B, T, self.heads, self.act, self.causal)     # one-time setup inside the launchers
            if len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            pool = torch.cuda.MemPool()
            plan = hip.Plan()
            with torch.cuda.use_mem_pool(pool, device=h.device):
                buf = torch.empty_like(h)
                ops.copy_rows(buf, h)
                with plan.record():
                    out = _encoder_forward(self.layers, buf, B, T, self.heads, self.act, self.causal)
            self._plans[key] = (plan, buf, out, pool)
            return out.clone()
        plan, buf, out, _ = e
        ops.copy_rows(buf, h)
        plan.replay()
        return out.clone()        # (the plan's output buffer is rewritten by the next request)


def _act_code(name: str) -> int:
    return ops.ACT_QUICK_GELU if name == 'quick_gelu' else ops.ACT_GELU


class _TextModel:
    def __init__(self, sd, cfg, dev):
        t = cfg.text
        self.cfg, self.dev = t, dev
        self.tok = ops.f16(sd['text_model.embeddings.token_embedding.weight'], dev)
        self.pos = ops.f16(sd['text_model.embeddings.position_embedding.weight'], dev)
        self.layers = [_Layer(sd, f'text_model.encoder.layers.{i}', dev)
                       for i in range(t.num_hidden_layers)]
        self.fln = (ops.f32(sd['text_model.final_layer_norm.weight'], dev),
                    ops.f32(sd['text_model.final_layer_norm.bias'], dev))
        self._stack = _PlannedEncoder(self.layers, t.num_attention_heads, _act_code(t.hidden_act), True)

    def __call__(self, input_ids: torch.Tensor, **_):
        hip.require_device(input_ids)
        ids = input_ids.to(torch.int64).contiguous()
        B, L = ids.shape
        D = self.cfg.hidden_size
        h = torch.empty((B * L, D), dtype=torch.float16, device=ids.device)
        hip.call('fd_embed_tokens_f16', ids.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(),
                 h.data_ptr(), B, L, D, self.cfg.vocab_size, hip.stream())
        h = self._stack(h, B, L)
        out = ops.layernorm(h, *self.fln, out_f32=True)
        return (out.view(B, L, D),)


class _VisionModel:
    def __init__(self, sd, cfg, dev):
        v = cfg.vision
        self.cfg, self.dev = v, dev
        self.patch = ops.prep_conv(sd['vision_model.embeddings.patch_embedding.weight'], None, dev)
        self.cls = ops.f16(sd['vision_model.embeddings.class_embedding'], dev)
        self.pos = ops.f16(sd['vision_model.embeddings.position_embedding.weight'], dev)
        self.pre = (ops.f32(sd['vision_model.pre_layrnorm.weight'], dev),
                    ops.f32(sd['vision_model.pre_layrnorm.bias'], dev))
        self.post = (ops.f32(sd['vision_model.post_layernorm.weight'], dev),
                     ops.f32(sd['vision_model.post_layernorm.bias'], dev))
        self.layers = [_Layer(sd, f'vision_model.encoder.layers.{i}', dev)
                       for i in range(v.num_hidden_layers)]
        self._stack = _PlannedEncoder(self.layers, v.num_attention_heads, _act_code(v.hidden_act), False)

    def embeddings(self, pixel_values: torch.Tensor) -> torch.Tensor:
        '''(B,3,224,224) fp32 -> (B,257,C) fp16: patch conv (im2col + MFMA GEMM), class token,
        position embeddings.'''
        hip.require_device(pixel_values)
        v = self.cfg
        x = ops.nchw_to_nhwc(pixel_values)
        p = ops.conv2d(x, self.patch, stride=v.patch_size, pad=(0, 0))
        T = p.HW + 1
        out = torch.empty((x.B, T, v.hidden_size), dtype=torch.float16, device=x.t.device)
        hip.call('fd_vit_assemble_f16', p.t.data_ptr(), self.cls.data_ptr(), self.pos.data_ptr(),
                 out.data_ptr(), x.B, T, v.hidden_size, hip.stream())
        return out

    def _ln(self, h: torch.Tensor, wb) -> torch.Tensor:
        B, T, C = h.shape
        return ops.layernorm(h.reshape(B * T, C), *wb).view(B, T, C)

    def pre_layrnorm(self, h):
        return self._ln(h, self.pre)

    def post_layernorm(self, h):
        return self._ln(h, self.post)

    def encoder(self, inputs_embeds: torch.Tensor, **_):
        B, T, C = inputs_embeds.shape
        h = self._stack(inputs_embeds.reshape(B * T, C).contiguous(), B, T)
        return (h.view(B, T, C),)


class CLIPModel():
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: CLIPConfig = CLIP_VIT_L14,
                 device='cuda'):
        hip.lib()
        self.cfg = config
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('flexdiffuse_amd.CLIPModel needs a HIP device (no CPU fallback)')
        missing = [k for k in clip_param_shapes(config) if k not in state_dict
                   and k not in ('logit_scale', 'text_projection.weight')]
        if missing:
            raise KeyError(f'CLIP state dict is missing {len(missing)} keys, e.g. {missing[:3]}')
        self.text_model = _TextModel(state_dict, config, self.device)
        self.vision_model = _VisionModel(state_dict, config, self.device)
        self._vproj = ops.prep_linear(state_dict['visual_projection.weight'], None, self.device)

    def to(self, device):
        return self

    def visual_projection(self, h: torch.Tensor) -> torch.Tensor:
        '''(B,T,C) fp16 -> (B,T,projection_dim) fp32 (no bias).'''
        B, T, C = h.shape
        out = ops.gemm(h.reshape(B * T, C), self._vproj, out_f32=True)
        return out[:, :self.cfg.projection_dim].reshape(B, T, self.cfg.projection_dim)
