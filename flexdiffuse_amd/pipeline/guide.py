'''Noise-prediction guides -- host-side mirror of the reference's `pipeline/guide.py`
(GuideBase :8-36, SimpleGuide :39-64, PromptGuide :67-72): same attributes
(`batch_size, steps, guidance, uncond_embeds, encoder, unet`) and the same
`noise_pred(latents, step)` protocol, so custom guides written against the reference keep
working.  `SimpleGuide` additionally exposes what FlexPipeline's fused device loop needs.
'''
from __future__ import annotations

from typing import List, Union

import torch

from .. import ops


class GuideBase():
    def __init__(self, encoder, unet, guidance: float, steps: int) -> None:
        '''Args mirror pipeline/guide.py:9-29 (encoder: CLIPEncoder, unet, guidance scale as in
        classifier-free guidance -- enabled when > 1 --, number of denoising steps).'''
        self.encoder = encoder
        self.unet = unet
        self.uncond_embeds = encoder.prompt('')
        self.batch_size = 1
        self.guidance = guidance
        self.steps = steps

    def noise_pred(self, latents: torch.Tensor, step: int) -> torch.Tensor:
        raise NotImplementedError('noise_pred must be implemented.')


class SimpleGuide(GuideBase):
    def __init__(self, encoder, unet, guidance: float, steps: int, clip_embeds: torch.Tensor):
        GuideBase.__init__(self, encoder, unet, guidance, steps)
        self.embeds = clip_embeds
        self.batch_size = self.embeds.shape[0]
        self._stack = None

    @property
    def classifier_free_guidance(self) -> bool:
        return self.guidance > 1.0

    def stacked_embeds(self) -> torch.Tensor:
        '''[uncond]*B + embeds (pipeline/guide.py:49-53), built once instead of every step so
        the UNet's cross-attention K/V projections of the context are computed once.'''
        if not self.classifier_free_guidance:
            return self.embeds
        if self._stack is None or self._stack_src is not self.embeds:
            B = self.batch_size
            self._stack = torch.cat([self.uncond_embeds.to(self.embeds.dtype).expand(B, -1, -1),
                                     self.embeds]).contiguous()
            self._stack_src = self.embeds
        return self._stack

    def noise_pred(self, latents: torch.Tensor, step) -> torch.Tensor:
        cfg = self.classifier_free_guidance
        B, C, H, W = latents.shape
        # one UNet pass over [uncond | cond]; latents are duplicated inside the layout kernel
        eps = self.unet.forward_nhwc(latents, step, self.stacked_embeds(), rep=2 if cfg else 1)
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=latents.device)
        # u + g (t - u)  (pipeline/guide.py:59-63), NHWC fp32 -> NCHW fp32
        ops.cfg_ddim_step(None, eps, B, C, H * W, cfg, self.guidance, do_step=False, eps_out=out)
        return out


class PromptGuide(SimpleGuide):
    def __init__(self, encoder, unet, guidance: float, steps: int, prompt: Union[str, List[str]]):
        SimpleGuide.__init__(self, encoder, unet, guidance, steps, encoder.prompt(prompt))
        self.prompt = prompt
