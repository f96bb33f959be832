'''Prompt / image guided embeddings and the tween between them -- host-side mirror of
the reference's `guidance.py` (same names, argument meaning and error behaviour), with
the arithmetic done by the gfx950 kernels of csrc/guidance.hip:

  map_emb          <- guidance.py:23-85    fd_guidance_map
  Tweener.tween    <- guidance.py:196-272  fd_guidance_tween (B prompts per launch)
  ConceptMapper    <- guidance.py:275-312  fd_guidance_map x2 + fd_guidance_concept_override
  Guide.embeds     <- guidance.py:315-474

Differences from the reference, all deliberate:
  * a batch of prompts with a guide works (the reference raises IndexError, E2): every
    prompt is tweened against the guide exactly as a single prompt would be;
  * nothing is printed unless `verbose=True` (printing similarities forces a device
    sync); the blend weights of the last call are kept in `Tweener.last_weights`;
  * adjacent equal similarity peaks raise ZeroDivisionError like the reference (E3)
    when `strict=True` (default; costs one 4-byte read-back per call).
'''
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import hip
from .encode.clip import CLIPEncoder

CLIP_IMAGE_SIZE = 224
MAX_SINGLE_DIM = 512  # for stable diffusion image

GUIDE_ORDER_TEXT = 0
GUIDE_ORDER_ALIGN = 1
GUIDE_ORDER_DIRECT = 2


def _prep(t: torch.Tensor) -> torch.Tensor:
    hip.require_device(t)
    if t.dim() == 2:
        t = t[None]
    return t.to(torch.float32).contiguous()


def map_emb_device(alt_emb: torch.Tensor, txt_emb: torch.Tensor, alt_emb_reuse: bool = True,
                   guide_order: int = GUIDE_ORDER_ALIGN) -> Tuple[torch.Tensor, torch.Tensor]:
    '''Device form of `_map_emb`: (idx int32 (B,L), s float32 (B,L)), no host sync.'''
    alt, txt = _prep(alt_emb), _prep(txt_emb)
    B, L, D = txt.shape
    Ba, N, Da = alt.shape
    if Da != D or Ba not in (1, B):
        raise ValueError(f'guide tokens {tuple(alt.shape)} do not match text {tuple(txt.shape)}')
    ws = torch.empty(B * N * L, dtype=torch.float32, device=txt.device)
    idx = torch.empty((B, L), dtype=torch.int32, device=txt.device)
    s = torch.empty((B, L), dtype=torch.float32, device=txt.device)
    hip.call('fd_guidance_map', hip.ptr(alt), hip.ptr(txt), hip.ptr(ws), hip.ptr(idx),
             hip.ptr(s), B, int(Ba == B and B > 1), N, L, D, int(guide_order),
             int(bool(alt_emb_reuse)), hip.stream())
    return idx, s


def _map_emb(alt_emb: torch.Tensor, txt_emb: torch.Tensor, alt_emb_reuse: bool = True,
             guide_order: int = GUIDE_ORDER_ALIGN) -> np.ndarray:
    '''Drop-in for guidance.py:23-85: float64 ndarray (L, 2) of (guide index, alignment).'''
    idx, s = map_emb_device(alt_emb, txt_emb, alt_emb_reuse, guide_order)
    out = np.zeros((idx.shape[1], 2))
    out[:, 0] = idx[0].cpu().numpy()
    out[:, 1] = s[0].cpu().numpy().astype(np.float64)
    return out


class Tweener():
    def __init__(self,
                 threshold: Tuple[float, float] = (0.5, 0.5),
                 linear: Tuple[float, float] = (0.0, 0.5),
                 clustered: float = 0.5,
                 max_guidance: float = 0.5,
                 header_max: float = 0.15,
                 align_mode: int = GUIDE_ORDER_ALIGN,
                 mapping_reuse: bool = True,
                 strict: bool = True,
                 verbose: bool = False) -> None:
        self.threshold_floor = threshold[0]
        self.threshold_mult = threshold[1]
        self.linear_start = linear[0]
        self.linear_end = linear[1]
        self.clustered = clustered
        self.max_guidance = max_guidance
        self.header_max = header_max
        self.align_mode = align_mode
        self.mapping_reuse = mapping_reuse
        self.strict = strict
        self.verbose = verbose
        self.last_weights: Optional[torch.Tensor] = None
        self.last_map: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.last_status: Optional[torch.Tensor] = None

    def _params(self) -> hip.fd_tween_params:
        return hip.fd_tween_params(float(self.threshold_floor), float(self.threshold_mult),
                                   float(self.clustered), float(self.max_guidance),
                                   float(self.header_max), int(self.align_mode),
                                   int(bool(self.mapping_reuse)))

    def tween(self, base_emb: torch.Tensor, alt_emb: torch.Tensor) -> torch.Tensor:
        '''Tween B text embeddings (B,L,D) towards the guide tokens (1|B,N,D).'''
        base, alt = _prep(base_emb), _prep(alt_emb)
        B, L, D = base.shape
        Ba, N, Da = alt.shape
        if Da != D or Ba not in (1, B):
            raise ValueError(f'guide tokens {tuple(alt.shape)} do not match text '
                             f'{tuple(base.shape)}')
        dev = base.device
        # guidance.py:231-233: the linear ramp is produced by torch.linspace on the host
        lin_w = torch.linspace(self.linear_start, self.linear_end, steps=L).to(dev)
        ws = torch.empty(B * N * L, dtype=torch.float32, device=dev)
        out = torch.empty_like(base)
        weights = torch.empty((B, L), dtype=torch.float32, device=dev)
        idx = torch.empty((B, L), dtype=torch.int32, device=dev)
        s = torch.empty((B, L), dtype=torch.float32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        p = self._params()
        hip.call('fd_guidance_tween', hip.ptr(base), hip.ptr(alt), hip.ptr(lin_w), hip.ptr(ws),
                 hip.ptr(out), hip.ptr(weights), hip.ptr(idx), hip.ptr(s), hip.ptr(status),
                 B, int(Ba == B and B > 1), N, L, D, p, hip.stream())
        self.last_weights, self.last_map, self.last_status = weights, (idx, s), status
        if self.strict and bool((status != 0).any().item()):
            raise ZeroDivisionError('float division by zero (adjacent equal similarity '
                                    'peaks in clustered guidance)')
        if self.verbose:
            avg = s.double().mean(dim=1)
            for b in range(B):
                print(f'Tweening with, Avg Similarity: {avg[b].item():.2%}, '
                      f'Threshold: {self.threshold_floor:.2%}, '
                      f'Threshold Multiplier: {self.threshold_mult:.2%}, '
                      f'Clustered: {self.clustered:.2%}, '
                      f'Linear: {self.linear_start:.2%}-{self.linear_end:.2%}, '
                      f'Guidance Max: {self.max_guidance:.2%}')
                print('Alt Embed Blend Weights:', weights[b].shape, ':', weights[b].cpu())
        return out


class ConceptMapper():
    def __init__(self, guide_embeddings: torch.Tensor, concept_embeddings: torch.Tensor,
                 verbose: bool = False) -> None:
        self.guide_embeddings = _prep(guide_embeddings)
        self.concept_embeddings = _prep(concept_embeddings)
        self.verbose = verbose
        # guidance.py:280: concept prompt <-> image, each image token used once, text order
        self.concept_mappings = map_emb_device(self.guide_embeddings, self.concept_embeddings,
                                               False, GUIDE_ORDER_TEXT)

    def map(self, base_embeddings: torch.Tensor,
            output_embeddings: Optional[torch.Tensor] = None) -> torch.Tensor:
        base = _prep(base_embeddings)
        if output_embeddings is None:
            output_embeddings = base.clone()
        out = _prep(output_embeddings)
        if out.data_ptr() != output_embeddings.data_ptr():
            out = out.clone()
        B, L, D = base.shape
        N = self.guide_embeddings.shape[1]
        for b in range(B):
            # guidance.py:293: base <-> concept, reuse allowed, alignment order
            ct_idx, ct_s = map_emb_device(self.concept_embeddings, base[b:b + 1], True,
                                          GUIDE_ORDER_ALIGN)
            hip.call('fd_guidance_concept_override', hip.ptr(self.guide_embeddings),
                     hip.ptr(self.concept_mappings[0]), hip.ptr(ct_idx), hip.ptr(ct_s),
                     hip.ptr(out[b]), N, L, D, hip.stream())
        return out


class Guide():
    def __init__(self, clip, tokenizer, device: str = 'cuda', verbose: bool = False) -> None:
        '''Context for prompt / image embeddings and their tween (guidance.py:316-335).

        Args:
            clip: CLIP model container (flexdiffuse_amd.clip.CLIPModel or any object with the
                reference's duck-typed surface).
            tokenizer: tokenizer with `model_max_length` and the HF call signature.
            device: HIP device string.
        '''
        self.clip = clip
        self.tokenizer = tokenizer
        self.device = device
        self.verbose = verbose
        self.encoder = CLIPEncoder(clip, tokenizer)
        # header token of the placeholder prompt, used by pure image guidance
        self.placeholder_embed = self.encoder.prompt('{}')

    def embeds(self,
               prompt: Union[str, List[str]] = '',
               guide=None,
               mapping_concepts: str = '',
               guide_threshold_mult: float = 0.5,
               guide_threshold_floor: float = 0.5,
               guide_clustered: float = 0.5,
               guide_linear: Tuple[float, float] = (0.0, 0.5),
               guide_max_guidance: float = 0.5,
               guide_header_max: float = 0.15,
               guide_mode: int = GUIDE_ORDER_ALIGN,
               guide_reuse: bool = True) -> torch.Tensor:
        '''CLIP embeddings (B,77,D) for Stable Diffusion from text, image, or a tween of
        both.  Same arguments, defaults and ValueErrors as guidance.py:337-474.'''
        if isinstance(prompt, str):
            prompt = prompt.strip()
        elif isinstance(prompt, list):
            prompt = [ss for ss in (s.strip() for s in prompt) if ss]
        else:
            raise ValueError(f'`prompt` has to be of type `str` '
                             f'or `list` but is {type(prompt)}')
        if not prompt and guide is None:
            raise ValueError('No prompt, or guide image provided.')

        text_embeddings: Optional[torch.Tensor] = None
        guide_embeddings: Optional[torch.Tensor] = None
        concept_mapper: Optional[ConceptMapper] = None
        if prompt:
            text_embeddings = self.encoder.prompt(prompt)
        if guide is not None:
            if isinstance(guide, str):
                guide = guide.strip()
                if guide:
                    guide_embeddings = self.encoder.prompt(guide)
            else:
                guide_embeddings = self.encoder.image(guide)
                if mapping_concepts:
                    concept_mapper = ConceptMapper(guide_embeddings,
                                                   self.encoder.prompt(mapping_concepts),
                                                   verbose=self.verbose)
        tweener = Tweener((guide_threshold_floor, guide_threshold_mult), guide_linear,
                          guide_clustered, guide_max_guidance, guide_header_max, guide_mode,
                          guide_reuse, verbose=self.verbose)
        self.last_tweener = tweener

        if text_embeddings is not None:
            if guide_embeddings is not None:
                # one launch for the whole batch of prompts (reference: per-row loop, E2)
                clip_embeddings = tweener.tween(text_embeddings.float(), guide_embeddings.float())
                if concept_mapper is not None:
                    clip_embeddings = concept_mapper.map(text_embeddings.float(), clip_embeddings)
            else:
                clip_embeddings = text_embeddings
        else:
            assert guide_embeddings is not None
            if isinstance(guide, str):
                if self.verbose:
                    print('Warning: using the guide like prompt.. just use prompt.')
                clip_embeddings = guide_embeddings
            else:
                if self.verbose:
                    print('Warning: trying to guide purely from image, '
                          'this will generate weird stuff, enjoy :)')
                L = self.tokenizer.model_max_length
                clip_embeddings = guide_embeddings[:, :L, :].float().contiguous()
                hdr = self.placeholder_embed[0, 0, :].float().contiguous()
                B, L_, D = clip_embeddings.shape
                # guidance.py:469-472: move the header 85% towards the text header
                hip.call('fd_guidance_header_pull', hip.ptr(clip_embeddings), hip.ptr(hdr),
                         B, L_, D, hip.stream())
        return clip_embeddings
