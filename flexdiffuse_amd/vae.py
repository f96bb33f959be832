'''Stable-Diffusion VAE (AutoencoderKL) on gfx950 -- stands where the reference passes
diffusers' `AutoencoderKL` (call sites pipeline/flex.py:118 `vae.decode(z).sample`,
pipeline/flex.py:189-191 `vae.encode(x).latent_dist.sample(generator=)`).

Built from the same HIP kernels as the UNet (implicit-GEMM conv3x3 with the nearest-2x
upsample fused into the gather, GroupNorm+SiLU); the single-head d=512 mid-block attention
runs as two batched MFMA GEMMs around a row softmax.
'''
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import hip, ops
from .ops import Act
from .weights import SD_VAE, VAEConfig, vae_param_shapes


class _VRes:
    def __init__(self, sd, name, dev):
        self.n1g, self.n1b = ops.f32(sd[name + '.norm1.weight'], dev), ops.f32(sd[name + '.norm1.bias'], dev)
        self.n2g, self.n2b = ops.f32(sd[name + '.norm2.weight'], dev), ops.f32(sd[name + '.norm2.bias'], dev)
        self.conv1 = ops.prep_conv(sd[name + '.conv1.weight'], sd[name + '.conv1.bias'], dev)
        self.conv2 = ops.prep_conv(sd[name + '.conv2.weight'], sd[name + '.conv2.bias'], dev)
        self.short = None
        if name + '.conv_shortcut.weight' in sd:
            w = sd[name + '.conv_shortcut.weight']
            self.short = ops.prep_linear(w.reshape(w.shape[0], w.shape[1]),
                                         sd[name + '.conv_shortcut.bias'], dev)


class _VAttn:
    def __init__(self, sd, name, dev):
        self.g, self.b = ops.f32(sd[name + '.group_norm.weight'], dev), ops.f32(sd[name + '.group_norm.bias'], dev)
        lin = lambda n: ops.prep_linear(sd[f'{name}.{n}.weight'], sd[f'{name}.{n}.bias'], dev)
        self.q, self.k, self.v, self.o = lin('query'), lin('key'), lin('value'), lin('proj_attn')


class DiagonalGaussian():
    '''`latent_dist` of `encode()`: mean/logvar NCHW fp32 on device.'''
    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor):
        self.mean, self.logvar = mean, logvar

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        # noise is drawn on the HOST generator and uploaded (parity and
        # shard invariance need a CPU stream of random numbers)
        gen_dev = getattr(generator, 'device', torch.device('cpu'))
        noise = torch.randn(self.mean.shape, generator=generator, device=gen_dev,
                            dtype=torch.float32).to(self.mean.device)
        return self.sample_with(noise)

    def sample_with(self, noise: torch.Tensor) -> torch.Tensor:
        std_noise = ops.axpby(self.logvar, noise, 0.0, 1.0, exp_half_x=True)
        return ops.axpby(self.mean, std_noise, 1.0, 1.0)

    def mode(self):
        return self.mean


class AutoencoderKL():
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: VAEConfig = SD_VAE,
                 device='cuda', encoder: bool = True):
        hip.lib()
        self.cfg = config
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('flexdiffuse_amd.AutoencoderKL needs a HIP device (no CPU fallback)')
        sd, dev, cfg = state_dict, self.device, config
        needed = [k for k in vae_param_shapes(cfg) if encoder or not k.startswith(('encoder.', 'quant_conv'))]
        missing = [k for k in needed if k not in sd]
        if missing:
            raise KeyError(f'VAE state dict is missing {len(missing)} keys, e.g. {missing[:3]}')
        self.G = cfg.norm_num_groups
        ch = cfg.block_out_channels
        rev = list(reversed(ch))
        L = cfg.latent_channels
        conv = lambda n, **kw: ops.prep_conv(sd[n + '.weight'], sd[n + '.bias'], dev, **kw)
        # ---- decoder ----
        w = torch.zeros((8, 8)); w[:L, :L] = sd['post_quant_conv.weight'].reshape(L, L)
        b = torch.zeros(8); b[:L] = sd['post_quant_conv.bias']
        self.post_quant = ops.prep_linear(w, b, dev)
        self.d_conv_in = conv('decoder.conv_in', cin_pad=8)
        self.d_mid = (_VRes(sd, 'decoder.mid_block.resnets.0', dev),
                      _VAttn(sd, 'decoder.mid_block.attentions.0', dev),
                      _VRes(sd, 'decoder.mid_block.resnets.1', dev))
        self.d_up = []
        for i in range(len(rev)):
            res = [_VRes(sd, f'decoder.up_blocks.{i}.resnets.{j}', dev)
                   for j in range(cfg.layers_per_block + 1)]
            up = conv(f'decoder.up_blocks.{i}.upsamplers.0.conv') if i != len(rev) - 1 else None
            if up is not None and up.cin % 64 == 0 and not up.im2col:
                # Upsample2D as four 2x2 parity convolutions of the low-resolution map (4/9 of the MACs)
                n = f'decoder.up_blocks.{i}.upsamplers.0.conv'
                up = (up, ops.prep_conv_up_phases(sd[n + '.weight'], sd[n + '.bias'], dev))
            elif up is not None:
                up = (up, None)
            self.d_up.append((res, up))
        self.d_out_g, self.d_out_b = ops.f32(sd['decoder.conv_norm_out.weight'], dev), \
            ops.f32(sd['decoder.conv_norm_out.bias'], dev)
        self.d_conv_out = conv('decoder.conv_out')
        # ---- encoder (img2img) ----
        self.has_encoder = encoder
        if encoder:
            self.e_conv_in = conv('encoder.conv_in', cin_pad=4)
            self.e_down = []
            for i in range(len(ch)):
                res = [_VRes(sd, f'encoder.down_blocks.{i}.resnets.{j}', dev)
                       for j in range(cfg.layers_per_block)]
                dn = conv(f'encoder.down_blocks.{i}.downsamplers.0.conv') if i != len(ch) - 1 else None
                self.e_down.append((res, dn))
            self.e_mid = (_VRes(sd, 'encoder.mid_block.resnets.0', dev),
                          _VAttn(sd, 'encoder.mid_block.attentions.0', dev),
                          _VRes(sd, 'encoder.mid_block.resnets.1', dev))
            self.e_out_g, self.e_out_b = ops.f32(sd['encoder.conv_norm_out.weight'], dev), \
                ops.f32(sd['encoder.conv_norm_out.bias'], dev)
            self.e_conv_out = conv('encoder.conv_out')
            w = sd['quant_conv.weight']
            self.quant = ops.prep_linear(w.reshape(w.shape[0], w.shape[1]), sd['quant_conv.bias'], dev)

    def to(self, device):
        return self

    def _res(self, r: _VRes, x: Act) -> Act:
        h = ops.groupnorm(x, r.n1g, r.n1b, self.G, 1e-6, True)
        h = ops.conv2d(h, r.conv1)
        h = ops.groupnorm(h, r.n2g, r.n2b, self.G, 1e-6, True)
        sc = x.t if r.short is None else ops.gemm(x.t, r.short)
        return ops.conv2d(h, r.conv2, residual=sc)

    def _attn(self, a: _VAttn, x: Act) -> Act:
        B, N, C = x.B, x.HW, x.C
        h = ops.groupnorm(x, a.g, a.b, self.G, 1e-6, False)
        q, k = ops.gemm(h.t, a.q), ops.gemm(h.t, a.k)
        vt = ops.gemm_vt(h.t, a.v, B, N, (N + 7) // 8 * 8)
        # softmax((q C^-1/4)(k C^-1/4)^T) v  ==  softmax(q k^T / sqrt(C)) v
        s = ops.bgemm(q.view(B, N, C), k.view(B, N, C), alpha=C ** -0.5)
        ops.softmax_rows_(s)
        o = ops.bgemm(s, vt[:, :, :N])
        return Act(ops.gemm(o.view(B * N, C), a.o, residual=x.t), B, x.H, x.W)

    # ---- decode ---------------------------------------------------------------------------
    def decode_nhwc(self, z: torch.Tensor, scale: float = 1.0) -> Act:
        '''(B,4,h,w) fp32 latents (multiplied by `scale`) -> NHWC fp32 image Act [B*H*W][4].
        Large batches are decoded in sample chunks: the LDS-DMA GEMM tiles and GroupNorm address
        a tensor through 32-bit buffer offsets, so every activation must stay below 2 GiB -- the
        widest is the full-resolution map of the narrowest decoder level (128 channels for SD:
        32 images at 512x512, 14 at 768x768); the chunks keep it at <= 1 GiB.'''
        hip.require_device(z)
        B, _, hh, ww = z.shape
        up = 2 ** (len(self.cfg.block_out_channels) - 1)
        per_sample = hh * up * ww * up * self.cfg.block_out_channels[0]      # elements (fp16)
        chunk = max(1, getattr(self, 'decode_chunk_elems', 1 << 29) // max(per_sample, 1))
        if B > chunk:
            parts = [self.decode_nhwc(z[i:i + chunk], scale) for i in range(0, B, chunk)]
            return Act(torch.cat([p.t for p in parts]), B, parts[0].H, parts[0].W)
        x = ops.nchw_to_nhwc(z, c_pad=8, scale=scale)
        h = Act(ops.gemm(x.t, self.post_quant), x.B, x.H, x.W)
        h = ops.conv2d(h, self.d_conv_in)
        h = self._res(self.d_mid[0], h)
        h = self._attn(self.d_mid[1], h)
        h = self._res(self.d_mid[2], h)
        for res, up in self.d_up:
            for r in res:
                h = self._res(r, h)
            if up is not None:
                if up[1] is not None and ops.up_phases_supported(h.B * h.HW, up[0].cout, up[0].cin) and h.t.is_contiguous():
                    h = ops.conv2d_up_phases(h, up[1])
                else:
                    h = ops.conv2d(h, up[0], up=True)
        h = ops.groupnorm(h, self.d_out_g, self.d_out_b, self.G, 1e-6, True)
        return ops.conv2d(h, self.d_conv_out, out_f32=True)

    def decode(self, z: torch.Tensor):
        img = self.decode_nhwc(z)
        return SimpleNamespace(sample=ops.nhwc_to_nchw(img.t, img.B, self.cfg.out_channels, img.H, img.W))

    # ---- encode ---------------------------------------------------------------------------
    def encode(self, x: torch.Tensor):
        if not self.has_encoder:
            raise RuntimeError('this AutoencoderKL was built without its encoder')
        hip.require_device(x)
        h = ops.conv2d(ops.nchw_to_nhwc(x, c_pad=4), self.e_conv_in)
        for res, dn in self.e_down:
            for r in res:
                h = self._res(r, h)
            if dn is not None:   # diffusers: F.pad(x, (0,1,0,1)) then stride-2 conv, no padding
                h = ops.conv2d(h, dn, stride=2, pad=(0, 0), out_hw=(h.H // 2, h.W // 2))
        h = self._res(self.e_mid[0], h)
        h = self._attn(self.e_mid[1], h)
        h = self._res(self.e_mid[2], h)
        h = ops.groupnorm(h, self.e_out_g, self.e_out_b, self.G, 1e-6, True)
        h = ops.conv2d(h, self.e_conv_out)
        m = ops.gemm(h.t, self.quant, out_f32=True)
        L = self.cfg.latent_channels
        mom = ops.nhwc_to_nchw(m, h.B, 2 * L, h.H, h.W)
        mean, logvar = mom[:, :L].contiguous(), mom[:, L:].clamp(-30.0, 20.0).contiguous()
        return SimpleNamespace(latent_dist=DiagonalGaussian(mean, logvar))
