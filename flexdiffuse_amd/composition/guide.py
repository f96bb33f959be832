'''Region-composited guidance -- host-side mirror of the reference's `CompositeGuide`
(composition/guide.py:32-139) and `encode_schema` (composition/embeds.py:28-44): ONE UNet
batch over [uncond, background, entity_1..n] on the same latents, each entity's noise
prediction blended onto the background inside its latent-space rectangle
(bg + blend * (entity - bg)), then classifier-free guidance against the unconditional row.

Like the reference it is defined for batch_size == 1 (the reference concatenates
`latents` once per embedding row).  The style-blend embedding the reference computes at
composition/guide.py:114-121 is dead code there and is not evaluated here.
'''
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch

from .. import hip, ops
from ..pipeline.guide import GuideBase
from .schema import EntitySchema, Schema


def px_to_block(px_shape: Sequence[int]) -> Tuple[int, ...]:
    return tuple(px // 8 for px in px_shape)


@dataclass
class EntityEmbeds():
    embed: torch.Tensor
    offset_blocks: Tuple[int, int]
    size_blocks: Tuple[int, int]
    blend: float


class CompositeGuide(GuideBase):
    def __init__(self, encoder, unet, guidance: float, schema: Schema, steps: int,
                 batch_size: int = 1):
        GuideBase.__init__(self, encoder, unet, guidance, steps)
        if batch_size != 1:
            raise ValueError('CompositeGuide is defined for batch_size == 1 (as in the reference)')
        self.schema = schema
        self.background_embed = encoder.prompt(schema.background_prompt)
        self.entities: List[EntityEmbeds] = [
            EntityEmbeds(encoder.prompt(e.prompt), px_to_block(e.offset), px_to_block(e.size),
                         e.blend) for e in schema.entities]
        self.batch_size = batch_size
        self.classifier_free_guidance = self.guidance > 1.0
        rows = [self.background_embed] + [e.embed for e in self.entities]
        if self.classifier_free_guidance:
            rows = [self.uncond_embeds] * self.batch_size + rows
        self.embed_tensor = torch.cat([r.float() for r in rows]).contiguous()

    def noise_pred(self, latents: torch.Tensor, step) -> torch.Tensor:
        E = self.embed_tensor.shape[0]
        _, C, H, W = latents.shape
        eps = self.unet.forward_nhwc(latents, step, self.embed_tensor, rep=E)
        stack = ops.nhwc_to_nchw(eps, E, C, H, W)            # (E,C,H,W) fp32
        first = 1 if self.classifier_free_guidance else 0
        bg = stack[first]
        for k, e in enumerate(self.entities):
            (ow, oh), (sw, sh) = e.offset_blocks, e.size_blocks
            # composition/guide.py:86-98 slices noise[:, :, oh:oh+sh, ow:ow+sw]: Python slice
            # semantics -- a box past the canvas is clipped, a NEGATIVE start counts from the end
            # of the axis (usually leaving an empty box, i.e. no blend at all)
            y0, y1, _ = slice(oh, oh + sh).indices(H)
            x0, x1, _ = slice(ow, ow + sw).indices(W)
            if y1 <= y0 or x1 <= x0:
                continue
            hip.call('fd_region_blend_f32', bg.data_ptr(), stack[first + 1 + k].data_ptr(), C, H, W,
                     y0, x0, y1 - y0, x1 - x0, float(e.blend), hip.stream())
        if not self.classifier_free_guidance:
            return bg[None].contiguous()
        out = torch.empty((1, C, H, W), dtype=torch.float32, device=latents.device)
        # rows 0 (uncond) and 1 (composited background) are contiguous: u + g (bg - u)
        ops.cfg_ddim_step(None, stack[:2].reshape(-1, 1), C, 1, H * W, True, self.guidance,
                          do_step=False, eps_out=out)
        return out
