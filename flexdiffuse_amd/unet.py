'''Stable-Diffusion UNet on gfx950 -- the container the pipeline receives where the
reference receives diffusers' `UNet2DConditionModel` (call sites pipeline/guide.py:56-58,
pipeline/flex.py:101-102,224; SURVEY.md 8b "duck-typed model objects").

Same call surface: `unet(latents, t, encoder_hidden_states=ctx).sample`, `.in_channels`,
`.config[...]`, `.set_attention_slice(...)`.  Every op is a hand-written HIP kernel behind
the C ABI (ops.py): implicit-GEMM conv3x3 with fused bias / time-embedding / residual
epilogues, GroupNorm+SiLU, flash self/cross attention, fused GEGLU.  Activations are NHWC
fp16, accumulation fp32.  Step-invariant work is hoisted out of the denoising loop: the
cross-attention K / V^T projections of the text context are cached per context tensor, and
all 22 ResBlock time-embedding projections are one GEMM per step.
'''
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

from . import hip, ops
from .ops import Act
from .weights import SD15_UNET, UNetConfig, unet_param_shapes, unet_up_plan


class _Res:
    def __init__(self, sd, name, dev, temb_off):
        self.n1g, self.n1b = ops.f32(sd[name + '.norm1.weight'], dev), ops.f32(sd[name + '.norm1.bias'], dev)
        self.n2g, self.n2b = ops.f32(sd[name + '.norm2.weight'], dev), ops.f32(sd[name + '.norm2.bias'], dev)
        self.conv1 = ops.prep_conv(sd[name + '.conv1.weight'], sd[name + '.conv1.bias'], dev)
        self.conv2 = ops.prep_conv(sd[name + '.conv2.weight'], sd[name + '.conv2.bias'], dev)
        self.short = None
        self.sc_fused = False
        if name + '.conv_shortcut.weight' in sd:
            w = sd[name + '.conv_shortcut.weight']
            w2 = sd[name + '.conv2.weight']
            if os.environ.get('FD_UNET_SC_FUSE', '1') != '0' and w.shape[1] % 64 == 0 and w2.shape[1] % 64 == 0:
                # the 1x1 shortcut rides in conv2's own K loop (fd_gemm_desc.A2 / K2): no shortcut launch, no
                # shortcut tensor written and re-read as conv2's residual
                self.conv2 = ops.prep_conv_shortcut(w2, sd[name + '.conv2.bias'], w.reshape(w.shape[0], w.shape[1]),
                                                    sd[name + '.conv_shortcut.bias'], dev)
                self.sc_fused = True
            else:
                self.short = ops.prep_linear(w.reshape(w.shape[0], w.shape[1]),
                                             sd[name + '.conv_shortcut.bias'], dev)
        self.cout = self.conv1.cout
        self.temb_off = temb_off  # column offset into the fused time-embedding projection


class _Attn:
    def __init__(self, sd, name, dev, heads, linear_proj, groups=32):
        g = lambda k: sd[name + k]
        self.heads = heads
        self.ng, self.nb = ops.f32(g('.norm.weight'), dev), ops.f32(g('.norm.bias'), dev)
        w = g('.proj_in.weight')
        self.proj_in = ops.prep_linear(w.reshape(w.shape[0], w.shape[1]), g('.proj_in.bias'), dev)
        # GroupNorm fold: norm -> proj_in as ONE per-sample linear layer on the un-normalised input (one statistics pass,
        # no normalised activation written and re-read; ops.gn_fold_supported picks the levels where it pays)
        self.gnf = ops.prep_gn_fold(w, g('.proj_in.bias'), g('.norm.weight'), g('.norm.bias'), groups, 1e-6, dev)
        w = g('.proj_out.weight')
        self.proj_out = ops.prep_linear(w.reshape(w.shape[0], w.shape[1]), g('.proj_out.bias'), dev)
        tb = '.transformer_blocks.0'
        self.ln = [(ops.f32(g(f'{tb}.norm{i}.weight'), dev), ops.f32(g(f'{tb}.norm{i}.bias'), dev))
                   for i in (1, 2, 3)]
        # the softmax scale (and the base-2 conversion) is folded into the q projections, so the
        # attention kernel exponentiates K.Q^T directly (fd_attention_desc.q_prescaled)
        wq = g(f'{tb}.attn1.to_q.weight')
        d = wq.shape[0] // heads
        self.q_pre = ops.attention_accepts_prescaled(d)
        qs = ops.QK_LOG2E * d ** -0.5 if self.q_pre else 1.0
        # self-attention q and k share their input: one GEMM with the weights stacked along N
        # LayerNorm fold (default): the three LayerNorms of the block are not launched; their consumer GEMMs
        # (q|k, v, cross q, GEGLU) read the un-normalised hidden states with gain-folded weights and apply
        # (rstd, mean) per row in their epilogues (fd_gemm_desc.ln_stats; one statistics pass per LayerNorm).
        # FD_UNET_LN_FOLD=0 keeps the separate LayerNorm kernels (A/B).
        self.ln_fold = os.environ.get('FD_UNET_LN_FOLD', '1') != '0'
        self.ln_emit = os.environ.get('FD_UNET_LN_EMIT', '1') != '0'   # 0: separate statistics pass everywhere (A/B)
        wqk = torch.cat([wq.float() * qs, g(f'{tb}.attn1.to_k.weight').float()], 0)
        wq2 = g(f'{tb}.attn2.to_q.weight').float() * qs
        if self.ln_fold:
            (g1, b1), (g2, b2), (g3, b3) = [(g(f'{tb}.norm{i}.weight'), g(f'{tb}.norm{i}.bias')) for i in (1, 2, 3)]
            self.qk1 = ops.prep_linear_ln(wqk, None, g1, b1, dev)
            self.v1 = ops.prep_linear_ln(g(f'{tb}.attn1.to_v.weight'), None, g1, b1, dev)
            # ... and q | k | v stacked for the one-launch form (ops.gemm_qkv: the V columns are stored transposed by the same launch)
            self.qkv1 = ops.prep_linear_ln(torch.cat([wqk, g(f'{tb}.attn1.to_v.weight').float()], 0), None, g1, b1, dev) \
                if ops.QKV_MERGE and wq.shape[0] % 160 == 0 else None
            self.q2 = ops.prep_linear_ln(wq2, None, g2, b2, dev)
            self.ff1 = ops.prep_linear_ln(g(f'{tb}.ff.net.0.proj.weight'), g(f'{tb}.ff.net.0.proj.bias'), g3, b3, dev,
                                          geglu=True)
        else:
            # self-attention q and k share their input: one GEMM with the weights stacked along N
            self.qk1 = ops.prep_linear(wqk, None, dev)
            self.v1 = ops.prep_linear(g(f'{tb}.attn1.to_v.weight'), None, dev)
            self.qkv1 = None
            self.q2 = ops.prep_linear(wq2, None, dev)
            self.ff1 = ops.prep_geglu(g(f'{tb}.ff.net.0.proj.weight'), g(f'{tb}.ff.net.0.proj.bias'), dev)
        self.o1 = ops.prep_linear(g(f'{tb}.attn1.to_out.0.weight'), g(f'{tb}.attn1.to_out.0.bias'), dev)
        self.k2 = ops.prep_linear(g(f'{tb}.attn2.to_k.weight'), None, dev)
        self.v2 = ops.prep_linear(g(f'{tb}.attn2.to_v.weight'), None, dev)
        self.o2 = ops.prep_linear(g(f'{tb}.attn2.to_out.0.weight'), g(f'{tb}.attn2.to_out.0.bias'), dev)
        self.ff2 = ops.prep_linear(g(f'{tb}.ff.net.2.weight'), g(f'{tb}.ff.net.2.bias'), dev)
        self.C = self.q2.N
        # proj_out folded THROUGH the feed-forward output layer: proj_out(ff2(f) + h) = f (Wp W2)^T + h Wp^T + (bp + Wp b2)
        # -- one GEMM over [f | h] (fd_gemm_desc.A2 / K2) instead of two, the block's last hidden state never
        # goes to HBM.  Weight product in fp32, rounded to fp16 once.  FD_UNET_FF_FOLD=0: the two launches (A/B).
        self.ffp = None
        wp = g('.proj_out.weight')
        wp = wp.reshape(wp.shape[0], wp.shape[1]).float()
        w2, b2 = g(f'{tb}.ff.net.2.weight').float(), g(f'{tb}.ff.net.2.bias').float()
        if os.environ.get('FD_UNET_FF_FOLD', '1') != '0' and w2.shape[1] % 64 == 0 and wp.shape[1] % 64 == 0:
            self.ffp = ops.prep_linear(torch.cat([wp @ w2, wp], 1), g('.proj_out.bias').float() + wp @ b2, dev)
        self.ctx_kv = None  # (K [Be*L][C], V^T [Be][C][ldv]) of the cached text context
        self.ctx_img = None  # the same context packed for the fused q-projection + cross-attention kernel


class UNetOutput(SimpleNamespace):
    def __getitem__(self, k):
        return self.sample if k in (0, 'sample') else getattr(self, k)


class UNet2DConditionModel():
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: UNetConfig = SD15_UNET,
                 device='cuda'):
        hip.lib()
        self.cfg = config
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('flexdiffuse_amd.UNet2DConditionModel needs a HIP device '
                               '(no CPU fallback)')
        self.in_channels = config.in_channels
        self.config = {'attention_head_dim': config.num_heads[0], 'in_channels': config.in_channels,
                       'cross_attention_dim': config.cross_attention_dim,
                       'block_out_channels': config.block_out_channels}
        missing = [k for k in unet_param_shapes(config) if k not in state_dict]
        if missing:
            raise KeyError(f'UNet state dict is missing {len(missing)} keys, e.g. {missing[:3]}')
        sd, dev, cfg = state_dict, self.device, config
        ch = cfg.block_out_channels
        heads = dict(zip(ch, cfg.num_heads))
        self.G = cfg.norm_num_groups
        self.conv_in = ops.prep_conv(sd['conv_in.weight'], sd['conv_in.bias'], dev, cin_pad=8)
        self.conv_in_nw = ops.prep_conv_narrow(sd['conv_in.weight'], sd['conv_in.bias'], dev)    # the one-launch form (fd_conv3x3_narrow_f16)
        self.t1 = ops.prep_linear(sd['time_embedding.linear_1.weight'], sd['time_embedding.linear_1.bias'], dev)
        self.t2 = ops.prep_linear(sd['time_embedding.linear_2.weight'], sd['time_embedding.linear_2.bias'], dev)
        temb_w, temb_b = [], []
        self._toff = 0

        def res(name):
            r = _Res(sd, name, dev, self._toff)
            temb_w.append(sd[name + '.time_emb_proj.weight'])
            temb_b.append(sd[name + '.time_emb_proj.bias'])
            self._toff += r.cout
            return r

        attn = lambda name, c: _Attn(sd, name, dev, heads[c], cfg.use_linear_projection, cfg.norm_num_groups)
        self.down: List[dict] = []
        for i, c in enumerate(ch):
            blk = {'res': [], 'attn': [], 'down': None}
            for j in range(cfg.layers_per_block):
                blk['res'].append(res(f'down_blocks.{i}.resnets.{j}'))
                blk['attn'].append(attn(f'down_blocks.{i}.attentions.{j}', c) if cfg.cross_attn[i] else None)
            if i != len(ch) - 1:
                blk['down'] = ops.prep_conv(sd[f'down_blocks.{i}.downsamplers.0.conv.weight'],
                                            sd[f'down_blocks.{i}.downsamplers.0.conv.bias'], dev)
            self.down.append(blk)
        self.mid_res0 = res('mid_block.resnets.0')
        self.mid_attn = attn('mid_block.attentions.0', ch[-1])
        self.mid_res1 = res('mid_block.resnets.1')
        self.up: List[dict] = []
        for i, (skips, c, has_attn, has_up) in enumerate(unet_up_plan(cfg)):
            blk = {'res': [], 'attn': [], 'up': None}
            for j in range(len(skips)):
                blk['res'].append(res(f'up_blocks.{i}.resnets.{j}'))
                blk['attn'].append(attn(f'up_blocks.{i}.attentions.{j}', c) if has_attn else None)
            if has_up:
                blk['up'] = ops.prep_conv(sd[f'up_blocks.{i}.upsamplers.0.conv.weight'],
                                          sd[f'up_blocks.{i}.upsamplers.0.conv.bias'], dev)
                # the same layer as four 2x2 parity convolutions of the low-resolution input (4/9 of the MACs): used
                # where the low-resolution map has enough rows to fill the chip (ops.up_phases_supported)
                blk['up_ph'] = ops.prep_conv_up_phases(sd[f'up_blocks.{i}.upsamplers.0.conv.weight'],
                                                       sd[f'up_blocks.{i}.upsamplers.0.conv.bias'], dev) \
                    if blk['up'].cin % 64 == 0 and not blk['up'].im2col else None
            self.up.append(blk)
        self.out_g, self.out_b = ops.f32(sd['conv_norm_out.weight'], dev), ops.f32(sd['conv_norm_out.bias'], dev)
        self.conv_out = ops.prep_conv(sd['conv_out.weight'], sd['conv_out.bias'], dev)
        # all ResBlock time-embedding projections as ONE [sum(Cout)][temb] GEMM per step
        self.temb_all = ops.prep_linear(torch.cat(temb_w, 0), torch.cat(temb_b, 0), dev)
        self.temb_total = self._toff
        self._attn_layers = [a for blk in self.down for a in blk['attn'] if a] + [self.mid_attn] + \
                            [a for blk in self.up for a in blk['attn'] if a]
        self._ctx_key = None
        # bumped whenever set_context REALLOCATES the cached K / V^T buffers (instead of rewriting
        # them in place): a captured HIP graph of the forward holds the old addresses and must not
        # be replayed after that (FlexPipeline keys its graph cache on this counter)
        self.ctx_generation = 0
        self._cat_plan = self._plan_concats()

    def _plan_concats(self) -> List[Optional[int]]:
        '''The decoder concatenates its running tensor h (Ch channels) with the encoder skips
        (diffusers' torch.cat in the up blocks).  Instead of copying both halves, every skip is
        written by its producer straight into the right-hand columns of a [M][Ch + Cs] buffer and
        the decoder op that produces h writes the left-hand columns.  Returns, per skip in
        encoder order, the Ch of the decoder tensor it will meet (None: an ordinary tensor and a
        copying concat).  Every consumer of a skip takes a row stride -- GroupNorm, the residual /
        shortcut operands, and the implicit-GEMM loader of the downsample convolutions (pixel stride).'''
        skip_c, excluded = [self.conv_in.cout], set()
        for blk in self.down:
            for r in blk['res']:
                skip_c.append(r.cout)
            if blk['down'] is not None:
                if blk['down'].im2col:
                    excluded.add(len(skip_c) - 1)      # narrow inputs (Cin % 64 != 0): the explicit im2col wants contiguous rows
                skip_c.append(blk['down'].cout)
        plan: List[Optional[int]] = [None] * len(skip_c)
        c, i = self.mid_res1.cout, len(skip_c) - 1
        for blk in self.up:
            for r in blk['res']:
                if i not in excluded and r.conv1.cin == c + skip_c[i] and c % 8 == 0 and skip_c[i] % 8 == 0:
                    plan[i] = c
                c, i = r.cout, i - 1
            if blk['up'] is not None:
                c = blk['up'].cout
        return plan

    # ---- reference surface --------------------------------------------------------------
    def set_attention_slice(self, slice_size):
        '''No-op: attention is flash-style (scores never materialised), so slicing
        (pipeline/flex.py:85-110) has nothing to save.'''
        self.attention_slice = slice_size

    def to(self, device):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states=None, return_dict=True):
        return self.forward(sample, timestep, encoder_hidden_states)

    # ---- context (step-invariant) -------------------------------------------------------
    def set_context(self, ctx: torch.Tensor):
        '''Project the text context through every cross-attention to_k / to_v once.'''
        key = (ctx.data_ptr(), ctx._version, tuple(ctx.shape))
        if key == self._ctx_key:
            return
        hip.require_device(ctx)
        Be, L, D = ctx.shape
        c16 = ops.cast_f16(ctx.reshape(Be * L, D)) if ctx.dtype != torch.float16 \
            else ctx.reshape(Be * L, D).contiguous()
        ldv = (L + 7) // 8 * 8
        realloc = False
        for a in self._attn_layers:
            old = a.ctx_kv
            if old is not None and old[0].shape == (Be * L, a.C) and old[1].shape == (Be, a.C, ldv):
                # same shapes as before: project in place so that a captured HIP graph of the
                # UNet (which holds these addresses) sees the new context
                ops.gemm(c16, a.k2, out=old[0])
                ops.gemm_vt(c16, a.v2, Be, L, ldv, out=old[1])
            else:
                a.ctx_kv = (ops.gemm(c16, a.k2), ops.gemm_vt(c16, a.v2, Be, L, ldv), L)
                a.ctx_img = None
                realloc = True
            # 8 heads x 40 (the 64x64 level): K / V^T also packed in MFMA fragment order for fd_xattn_q_f16
            if a.ln_fold and a.q_pre and ops.xattn_supported(a.heads, a.C // a.heads, L, ops.xattn_row_tile(a.C // a.heads)):
                a.ctx_img = ops.xattn_pack_kv(a.ctx_kv[0], a.ctx_kv[1], Be, L, a.heads, a.C // a.heads, out=a.ctx_img)
            else:
                a.ctx_img = None
        if realloc:
            self.ctx_generation += 1
        self._ctx_key = key
        self._ctx_ref = ctx  # keep the tensor alive so its data_ptr cannot be recycled

    # ---- blocks -----------------------------------------------------------------------------
    @staticmethod
    def _qkv_level(M: int) -> bool:
        '''Row counts at which the one-launch q | k | v projection is used.  Measured per launch against the two launches it replaces
        (tools/ab_qkv.py, profiles/r06_session_ab.txt sec. 7): equal at 65536 / 32768 rows, 0.7 us faster at 4096, 11.6 us faster at 1024,
        5.8 us SLOWER at 16384 rows x 640 channels (there the q|k GEMM alone runs on a ping-pong tile) -- so not between 8192 and 16384 rows;
        a launch boundary less either way.  FD_UNET_QKV_MIN_ROWS / _MAX_ROWS override the window (A/B).'''
        lo, hi = os.environ.get('FD_UNET_QKV_MIN_ROWS'), os.environ.get('FD_UNET_QKV_MAX_ROWS')
        if lo is not None or hi is not None:
            return int(lo or 0) <= M <= int(hi or (1 << 30))
        return not (8192 <= M <= 16384)

    def _gn1(self, r: Optional[_Res]) -> Optional[ops.GNSpec]:
        '''norm1 + SiLU of ResBlock `r` as a spec its input's PRODUCER can take (ops.conv2d(..., gn=)).'''
        return None if r is None else ops.GNSpec(r.n1g, r.n1b, self.G, 1e-5, True)

    def _attn_gn(self, a: Optional[_Attn], x: Act):
        '''What the producer of a transformer block's input can prepare for the block's GroupNorm: the normalised tensor (a GNSpec
        for ops.conv2d(..., gn=)) where the block reads one, or -- where the GroupNorm is folded into proj_in
        (ops.gn_fold_supported: only its statistics are needed) -- the group count, for the producer's partial sums (gn_parts=).'''
        if a is None:
            return None
        if ops.gn_fold_supported(x.B, x.HW, a.C, a.C, self.G):
            return self.G
        return ops.GNSpec(a.ng, a.nb, self.G, 1e-6, False)

    def _res(self, r: _Res, x: Act, temb: torch.Tensor, out: Optional[torch.Tensor] = None, xn: Optional[Act] = None,
             next_gn=None):
        '''ResBlock -> (output, GroupNorm `next_gn` of the output / its partial sums (next_gn an int: the group count) / None).  `xn`: norm1 + SiLU of `x` when the producer of x
        already made it (the split-K finish of the previous convolution); `next_gn`: the normalisation the consumer of this
        block's output starts with, handed to conv2 the same way.'''
        if isinstance(xn, Act):
            h = xn
        else:       # (xn an ops.GNParts: the statistics come from x's producer, only the apply pass runs)
            h = ops.groupnorm(x, r.n1g, r.n1b, self.G, 1e-5, True, parts=xn)
        # conv1 feeds nothing but norm2 + SiLU: where the convolution is split over K (16x16 / 8x8 levels) the finish pass of
        # the split normalises the tile it has just summed (fd_gemm_desc.gn_out) and conv1's own output is never written
        _, h = ops.conv2d(h, r.conv1, bias2=temb[:, r.temb_off:r.temb_off + r.cout], ld_bias2=self.temb_total,
                          gn=ops.GNSpec(r.n2g, r.n2b, self.G, 1e-5, True), keep=False)
        kw = dict(a2=x.t) if r.sc_fused else dict(residual=x.t if r.short is None else ops.gemm(x.t, r.short))
        if next_gn is None:
            return ops.conv2d(h, r.conv2, out=out, **kw), None
        if isinstance(next_gn, int):
            return ops.conv2d(h, r.conv2, out=out, gn_parts=next_gn, **kw)
        return ops.conv2d(h, r.conv2, out=out, gn=next_gn, keep=True, **kw)

    def _attn(self, a: _Attn, x: Act, rep: int = 1, out: Optional[torch.Tensor] = None, xn=None, next_parts: int = 0):
        '''Transformer block.  rep > 1: `x` holds B samples that are shared by `rep` branches of
        the cached context (CFG: [uncond]*B + cond on the same latents).  Everything up to the
        cross-attention query is branch-independent and computed once; the output has rep*B
        samples.  `xn`: what the producer of x prepared for the block's input GroupNorm (_attn_gn): the normalised tensor, or the
        partial sums of the statistics (ops.GNParts) where the GroupNorm is folded into proj_in.  `next_parts` = G > 0: the block's
        output is read next by a GroupNorm over G groups -> (output, ops.GNParts or None): the last GEMM's epilogue writes the partial sums.'''
        B, HW, C = x.B, x.HW, a.C
        d = C // a.heads
        gn_fold = ops.gn_fold_supported(B, HW, C, C, self.G)
        h = None if gn_fold else (xn if isinstance(xn, Act) else ops.groupnorm(x, a.ng, a.nb, self.G, 1e-6, False))
        # LayerNorm fold: the GEMM that PRODUCES a LayerNorm input also writes its row statistics where one
        # tile spans the row (C == 320: the 64x64 level); elsewhere one read-only statistics pass
        # (C == 320: finished pairs; wider rows: raw partial sums per 160-column tile + a tiny finalise launch)
        emit = (a.ln_fold and a.ln_emit) and ops.can_emit_row_stats(B * HW, C, C)

        def mkst(rows, batch=1):
            '''statistics buffer of a producer GEMM of `batch` x `rows` output rows (None: separate statistics pass)'''
            k = emit and ops.can_emit_row_stats(rows, C, C)
            if not k:
                return None
            return torch.empty((rows * batch, 2) if k == 1 else (k, rows * batch, 2), dtype=torch.float32, device=x.t.device)

        def fin(st, hh, parts_ok=False):
            '''(rstd, -mean rstd) of the rows of hh: emitted by its producer, finalised from its partial sums, or
            from one read-only pass.  parts_ok: the consumer is an fd_gemm_f16 launch that finalises the partial slabs itself,
            tile by tile (fd_gemm_desc.ln_stats_parts) -- the slabs are handed over as they are, no finalise launch.'''
            if st is None:
                return ops.ln_row_stats(hh)
            if st.dim() == 2 or (parts_ok and ops.LN_PARTS and st.shape[0] in (2, 4, 8)):
                return st
            return ops.ln_finalize_stats(st, C)
        if gn_fold:
            wb, bb = ops.gn_fold_linear(x, a.gnf, parts=xn if isinstance(xn, ops.GNParts) else None)
            st = mkst(HW, B)
            h = ops.gemm_per_sample(x.t, wb, bb, B, HW, ln_stats_out=st)
        else:
            st = mkst(B * HW)
            h = ops.gemm(h.t, a.proj_in, ln_stats_out=st)
        if a.ln_fold:
            st = fin(st, h, parts_ok=True)
            if a.qkv1 is not None and ops.qkv_merge_supported(B * HW, C, HW) and self._qkv_level(B * HW):
                qk, vt = ops.gemm_qkv(h, a.qkv1, B, HW, st)
            else:
                qk = ops.gemm(h, a.qk1, ln_stats=st)
                vt = ops.gemm_vt(h, a.v1, B, HW, (HW + 7) // 8 * 8, ln_stats=st)
        else:
            n = ops.layernorm(h, *a.ln[0])
            qk = ops.gemm(n, a.qk1)
            vt = ops.gemm_vt(n, a.v1, B, HW, (HW + 7) // 8 * 8)
        q, k = qk[:, :C], qk[:, C:]
        o = ops.attention(q, k, vt, B, a.heads, HW, HW, d, q_prescaled=a.q_pre)
        st = mkst(B * HW)
        h = ops.gemm(o, a.o1, residual=h, ln_stats_out=st)
        kc, vtc, L = a.ctx_kv
        xt = x.t
        if a.ctx_img is not None and HW % ops.xattn_row_tile(d) == 0:
            # q projection + cross-attention in one launch (the query matrix never goes to HBM); `rep`
            # context replicas share the queries
            o = ops.xattn_q(h, a.q2, fin(st, h, parts_ok=True), a.ctx_img, HW, L, a.heads, d, n_rep=rep)
        else:
            if a.ln_fold:
                q2 = ops.gemm(h, a.q2, ln_stats=fin(st, h, parts_ok=True))
            else:
                q2 = ops.gemm(ops.layernorm(h, *a.ln[1]), a.q2)
            if rep == 1:
                o = ops.attention(q2, kc, vtc, B, a.heads, HW, L, d, q_prescaled=a.q_pre)
            else:
                o = torch.empty((rep * B * HW, C), dtype=torch.float16, device=q2.device)
                for r in range(rep):
                    ops.attention(q2, kc[r * B * L:(r + 1) * B * L], vtc[r * B:(r + 1) * B], B, a.heads,
                                  HW, L, d, q_prescaled=a.q_pre, out=o[r * B * HW:(r + 1) * B * HW])
        if rep > 1:
            # the fan-out of the shared prefix: h and the block input are only RESIDUALS from here on (of the out-projection and of the
            # block's last GEMM), which can read them modulo the prefix's rows (fd_gemm_desc.residual_rows) -- no replicas in HBM
            if not ops.residual_wrap_supported(B * HW, rep):
                h, xt = ops.repeat_rows(h, rep), ops.repeat_rows(xt, rep)
            B = rep * B
        st = mkst(B * HW)
        h = ops.gemm(o, a.o2, residual=h, ln_stats_out=st)
        if a.ln_fold:
            # (GEGLU at K <= 640 runs the persistent tile, which reads finished statistics only: the finalise launch stays there)
            f = ops.gemm(h, a.ff1, act=ops.ACT_GEGLU, ln_stats=fin(st, h, parts_ok=C >= 1280))
        else:
            f = ops.gemm(ops.layernorm(h, *a.ln[2]), a.ff1, act=ops.ACT_GEGLU)
        if a.ffp is not None:
            if next_parts:
                o, parts = ops.gemm(f, a.ffp, a2=h, residual=xt, out=out, rows_per_sample=HW, gn_parts=next_parts)
                return Act(o, B, x.H, x.W), parts
            return Act(ops.gemm(f, a.ffp, a2=h, residual=xt, out=out), B, x.H, x.W)
        h = ops.gemm(f, a.ff2, residual=h)
        res = Act(ops.gemm(h, a.proj_out, residual=xt, out=out), B, x.H, x.W)
        return (res, None) if next_parts else res

    # ---- forward ----------------------------------------------------------------------------
    def time_bias(self, timestep, B: int) -> torch.Tensor:
        '''[B][sum Cout] fp32: every ResBlock's Linear(SiLU(time_embedding(t))).'''
        dev = self.device
        if isinstance(timestep, torch.Tensor):
            # a one-element device tensor is read by every sample (stride 0): the denoising loop
            # refreshes that scalar between replays of the forward's launch plan
            t = timestep.to(dev, torch.float32).reshape(-1)
            if t.numel() not in (1, B):
                raise ValueError(f'timestep tensor has {t.numel()} elements for a batch of {B}')
            t = t.contiguous()
        else:
            t = torch.full((1,), float(timestep), dtype=torch.float32, device=dev)
        dim = self.cfg.block_out_channels[0]
        e = torch.empty((B, dim), dtype=torch.float16, device=dev)
        hip.call('fd_timestep_embedding_f16', t.data_ptr(), 0 if t.numel() == 1 else 1, e.data_ptr(), B, dim,
                 hip.stream())
        e = ops.gemm(e, self.t1, act=ops.ACT_SILU)
        e = ops.gemm(e, self.t2, act=ops.ACT_SILU)   # only SiLU(emb) is ever consumed
        return ops.gemm(e, self.temb_all, out_f32=True)

    def time_bias_table(self, timesteps) -> torch.Tensor:
        '''[len(timesteps)][sum Cout] fp32: time_bias of every timestep of a denoising loop in ONE pass (row i = what
        time_bias(timesteps[i], 1) gives: each row is its own dot products, the batch does not enter).  The time embedding depends on
        nothing but t, so the loop computes it once per request instead of once per step (FlexPipeline._unet_eps).'''
        t = torch.tensor([float(v) for v in timesteps], dtype=torch.float32, device=self.device)
        return self.time_bias(t, t.numel())

    def forward_nhwc(self, sample: torch.Tensor, timestep, ctx: torch.Tensor,
                     rep: int = 1, temb: Optional[torch.Tensor] = None) -> torch.Tensor:
        '''(B,4,h,w) fp32 latents (replicated `rep` times along batch, e.g. for CFG) ->
        noise prediction as NHWC fp32 [rep*B*h*w][4].  `temb` [rep*B][sum Cout] fp32: the ResBlocks' time-embedding biases
        already computed (time_bias / time_bias_table rows, one timestep for the whole batch); `timestep` is then not used.'''
        hip.require_device(sample, ctx)
        self.set_context(ctx)
        # The `rep` branches see the same latents and timestep, so everything before the first
        # cross-attention (conv_in, the first ResBlock, the first block's self-attention) is
        # computed once on B samples and fanned out there (bit-identical per sample).
        share = rep > 1 and (temb is not None or not isinstance(timestep, torch.Tensor) or timestep.numel() == 1)
        # conv_in straight from the fp32 NCHW latents (one launch, replicas of the skip tensor included) where the layer has the shape for it
        direct = ops.CONV_IN_DIRECT and self.conv_in_nw is not None and (share or rep == 1) and sample.shape[3] <= 1024
        x = None if direct else ops.nchw_to_nhwc(sample, rep=1 if share else rep, c_pad=self.conv_in.cin)
        xB, xH, xW = (sample.shape[0], sample.shape[2], sample.shape[3]) if direct else (x.B, x.H, x.W)
        Be = xB * (rep if share else 1)
        if ctx.shape[0] != Be:
            raise ValueError(f'encoder_hidden_states batch {ctx.shape[0]} != latent batch {Be}')
        if temb is None:
            temb = self.time_bias(timestep, Be)
        elif temb.shape != (Be, self.temb_total) or temb.dtype != torch.float32 or not temb.is_contiguous():
            raise ValueError(f'temb must be a contiguous fp32 [{Be}][{self.temb_total}] tensor, got {tuple(temb.shape)} {temb.dtype}')
        fan = rep if share else 1      # > 1 while h still holds the shared B samples
        plan = self._cat_plan
        skips: List[tuple] = []        # (Act of the skip, its concat buffer or None, Ch)

        def slot(rows: int, cs: int):
            '''Right-hand view of a fresh concat buffer for the next skip (None: plain tensor).'''
            ch = plan[len(skips)] if len(skips) < len(plan) else None
            if ch is None:
                return None, None, None
            buf = torch.empty((rows, ch + cs), dtype=torch.float16, device=self.device)
            return buf, buf[:, ch:], ch

        def push(act: Act, buf, ch):
            skips.append((act, buf, ch))
            return act

        def left(rows: int, c: int):
            '''Left-hand view of the buffer the next concat will use, for the op producing h.'''
            if skips and skips[-1][1] is not None and skips[-1][2] == c and skips[-1][1].shape[0] == rows:
                return skips[-1][1][:, :c]
            return None

        HW = xH * xW
        buf, view, ch = slot(Be * HW, self.conv_in.cout)
        if direct and fan > 1:
            rep_out = view if view is not None else torch.empty((Be * HW, self.conv_in.cout), dtype=torch.float16, device=self.device)
            h = ops.conv3x3_narrow(sample.to(torch.float32), self.conv_in_nw, out2=rep_out, rep2=fan)
            push(Act(rep_out, Be, xH, xW), buf if view is not None else None, ch if view is not None else None)
        elif direct:
            h = push(ops.conv3x3_narrow(sample.to(torch.float32), self.conv_in_nw, out=view), buf, ch)
        elif fan > 1:
            h = ops.conv2d(x, self.conv_in)
            if view is None:
                push(Act(ops.repeat_rows(h.t, fan), Be, h.H, h.W), None, None)
            else:
                ops.repeat_rows(h.t, fan, out=view)
                push(Act(view, Be, h.H, h.W), buf, ch)
        else:
            h = push(ops.conv2d(x, self.conv_in, out=view), buf, ch)
        # `pend`: norm1 + SiLU of h for the NEXT ResBlock, made by the split-K finish of h's producer (deep levels; None elsewhere)
        pend: Optional[Act] = None
        seq = [r for blk in self.down for r in blk['res']] + [self.mid_res0]      # ResBlocks in execution order
        nres = 0
        for blk in self.down:
            for r, a in zip(blk['res'], blk['attn']):
                rows = Be * h.H * h.W
                buf, view, ch = slot(rows, r.cout)
                nres += 1
                if a is not None:
                    h, hn = self._res(r, h, temb[:h.B], xn=pend, next_gn=self._attn_gn(a, h))
                    # (the next ResBlock of this level reads the transformer block's output directly: its norm1 statistics come
                    # from the block's last GEMM where that launch can write them -- the 64x64 level)
                    if r is not blk['res'][-1]:
                        h, pend = self._attn(a, h, fan, out=view, xn=hn, next_parts=self.G)
                    else:
                        h, pend = self._attn(a, h, fan, out=view, xn=hn), None
                elif fan > 1:
                    h, pend = self._res(r, h, temb[:h.B], xn=pend)
                    if view is None:
                        h = Act(ops.repeat_rows(h.t, fan), Be, h.H, h.W)
                    else:
                        ops.repeat_rows(h.t, fan, out=view)
                        h = Act(view, Be, h.H, h.W)
                else:
                    # the next ResBlock reads exactly this tensor (no attention in between): its norm1 rides in conv2's finish
                    nxt = seq[nres] if blk['down'] is None or r is not blk['res'][-1] else None
                    h, pend = self._res(r, h, temb[:h.B], out=view, xn=pend, next_gn=self._gn1(nxt))
                fan = 1
                push(h, buf, ch)
            if blk['down'] is not None:
                buf, view, ch = slot(Be * (h.H // 2) * (h.W // 2), blk['down'].cout)
                if blk['down'].im2col:
                    h, pend = ops.conv2d(h, blk['down'], stride=2, out=view), None
                else:
                    h, pend = ops.conv2d(h, blk['down'], stride=2, out=view, gn=self._gn1(seq[nres]), keep=True)
                push(h, buf, ch)
        h, hn = self._res(self.mid_res0, h, temb, xn=pend, next_gn=self._attn_gn(self.mid_attn, h))
        h = self._attn(self.mid_attn, h, xn=hn)
        h, _ = self._res(self.mid_res1, h, temb, out=left(h.B * h.HW, self.mid_res1.cout))
        out_parts = None
        for blk in self.up:
            n = len(blk['res'])
            for j, (r, a) in enumerate(zip(blk['res'], blk['attn'])):
                s, sbuf, sch = skips.pop()
                if sbuf is not None and h.t.data_ptr() == sbuf.data_ptr() and h.C == sch:
                    h = Act(sbuf, h.B, h.H, h.W)          # both halves are already in place
                else:
                    h = Act(ops.concat_channels(ops.contiguous_rows(h.t), ops.contiguous_rows(s.t)), h.B, h.H, h.W)
                # where the result of this sub-block goes: the next concat's left half, unless an
                # upsample conv (contiguous input) follows
                last = j == n - 1
                dst = None if (last and blk['up'] is not None) else left(h.B * h.HW, r.cout)
                if a is not None:
                    h, hn = self._res(r, h, temb, next_gn=self._attn_gn(a, h))
                    if last and blk['up'] is None:      # the very last block: conv_norm_out reads its output
                        h, out_parts = self._attn(a, h, out=dst, xn=hn, next_parts=self.G)
                    else:
                        h = self._attn(a, h, out=dst, xn=hn)
                else:
                    h, _ = self._res(r, h, temb, out=dst)
            if blk['up'] is not None:
                dst = left(h.B * h.HW * 4, blk['up'].cout)
                if blk.get('up_ph') is not None and h.t.is_contiguous() and ops.up_phases_supported(h.B * h.HW, blk['up'].cout, blk['up'].cin):
                    h = ops.conv2d_up_phases(h, blk['up_ph'], out=dst)
                else:
                    h = ops.conv2d(h, blk['up'], up=True, out=dst)
        h = ops.groupnorm(h, self.out_g, self.out_b, self.G, 1e-5, True, parts=out_parts)
        return ops.conv2d(h, self.conv_out, out_f32=True).t

    def forward(self, sample, timestep, encoder_hidden_states) -> UNetOutput:
        B, C, H, W = sample.shape
        eps = self.forward_nhwc(sample, timestep, encoder_hidden_states)
        return UNetOutput(sample=ops.nhwc_to_nchw(eps, B, self.cfg.out_channels, H, W))
