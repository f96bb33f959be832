'''CLIP text / image encoders and image preprocessing -- host-side mirror of the
reference's `encode/clip.py` (preprocess :15-39, CLIPEncoder.prompt :47-65,
CLIPEncoder.image :67-100).

The image chain is host work in the reference too (PIL LANCZOS resize, torchvision
centre-crop / antialiased bicubic resize / normalise on CPU tensors, then `.to(device)`);
it is kept on the host here with the same arithmetic, including the quirk that CLIP's
mean/std are applied to a [-1, 1] tensor (encode/clip.py:82-84).  The
towers themselves run in HIP (flexdiffuse_amd/clip.py).
'''
from __future__ import annotations

from typing import Any, List, Union

import numpy as np
import torch
import torch.nn.functional as F

CLIP_IMAGE_SIZE = 224
MAX_SINGLE_DIM = 512  # for stable diffusion image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def sd_size(width: int, height: int) -> tuple:
    '''(w, h) the reference resizes an image to: long side 512, short side scaled and
    floored to a multiple of 64 (encode/clip.py:24-33).'''
    if height > width:
        return (int(width / (height / MAX_SINGLE_DIM)) // 64) * 64, MAX_SINGLE_DIM
    if width > height:
        return MAX_SINGLE_DIM, (int(height / (width / MAX_SINGLE_DIM)) // 64) * 64
    return MAX_SINGLE_DIM, MAX_SINGLE_DIM


def preprocess(image: Any) -> torch.Tensor:
    '''PIL image -> float32 (1,3,H,W) in [-1,1] at Stable-Diffusion size.'''
    from PIL import Image as _Image
    lanczos = getattr(_Image, 'LANCZOS', None) or _Image.Resampling.LANCZOS
    w, h = sd_size(*image.size)
    rgb = image.resize((w, h), resample=lanczos).convert('RGB')
    arr = np.asarray(rgb).astype(np.float32) / 255.0
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)[None]))
    return 2.0 * t - 1.0


def clip_pixels(guide_tensor: torch.Tensor) -> torch.Tensor:
    '''(1,3,H,W) in [-1,1] -> (1,3,224,224) CLIP input (encode/clip.py:77-84): centre crop
    to the short side (torchvision offsets int(round((h-c)/2))), antialiased bicubic
    resize, then CLIP mean/std.'''
    h, w = guide_tensor.shape[-2:]
    c = min(h, w)
    top = int(round((h - c) / 2.0))
    left = int(round((w - c) / 2.0))
    x = guide_tensor[..., top:top + c, left:left + c]
    x = F.interpolate(x, size=(CLIP_IMAGE_SIZE, CLIP_IMAGE_SIZE), mode='bicubic',
                      align_corners=False, antialias=True)
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype).view(-1, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype).view(-1, 1, 1)
    return (x - mean) / std


class CLIPEncoder():
    def __init__(self, clip, token) -> None:
        self.clip = clip
        self.token = token

    def prompt(self, prompt: Union[str, List[str]]) -> torch.Tensor:
        '''Text -> (B, model_max_length, D) last hidden state of the CLIP text tower.'''
        text_input = self.token(prompt, padding='max_length',
                                max_length=self.token.model_max_length, truncation=True,
                                return_tensors='pt')
        return self.clip.text_model(text_input.input_ids.to(self.clip.device))[0]

    def image(self, image: Any) -> torch.Tensor:
        '''PIL image -> (1, 257, D): every ViT token through post-layernorm and the visual
        projection (encode/clip.py:86-100).'''
        pixels = clip_pixels(preprocess(image)).to(self.clip.device)
        vm = self.clip.vision_model
        hidden = vm.pre_layrnorm(vm.embeddings(pixels))
        hidden = vm.encoder(inputs_embeds=hidden, output_attentions=False,
                            output_hidden_states=False, return_dict=True)[0]
        return self.clip.visual_projection(vm.post_layernorm(hidden[:, :, :]))
