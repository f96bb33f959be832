'''Harness -- host-side mirror of the reference's `utils.Runner` call recipe
(utils.py:54-166: seed handling :78-83, sequential sample batches :90, argument plumbing of
`gen` :114-166) and `image_grid` (:36-50).

`Runner.compose` (utils.py:168-207) is the caller of CompositeGuide: entity rows are parsed with
the reference's forgiving try/except, empty prompts dropped, a `Schema` built and run.

Out of scope here: model download (`from_pretrained`), PNG / grid files
and the filename scheme -- `Runner` takes state dicts (or builds seeded synthetic ones) and
returns the images.  Differences from the reference, all deliberate:
E5 `eta` stays 0.0 (the reference overwrites it with wall-clock seconds after the first
batch); E6 the generator is a CPU generator, so a seed gives the same images on any number
of GPUs; no CUDA autocast context is needed (the kernels are fp16 MFMA by construction).
'''
from __future__ import annotations

import gc
import math
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from . import build
from .encode.clip import CLIPEncoder
from .guidance import Guide
from .composition import CompositeGuide, EntitySchema, Schema
from .pipeline.guide import GuideBase, SimpleGuide

MAX_SEED = 2147483647          # the reference clamps seeds to int32 (utils.py:22)

# the keyword arguments `gen` forwards untouched to `Guide.embeds` (utils.py:151-162)
_EMBEDS_KEYS = ('mapping_concepts', 'guide_threshold_mult', 'guide_threshold_floor', 'guide_clustered',
                'guide_linear', 'guide_max_guidance', 'guide_header_max', 'guide_mode', 'guide_reuse')


def image_grid(imgs: Sequence[Any]):
    '''Paste PIL images into one sheet, row-major: ceil(sqrt(n)) columns and n // columns rows, so
    the last images are dropped when they do not fill a row (the reference's arithmetic,
    utils.py:36-50).'''
    from PIL import Image
    cols = math.ceil(math.sqrt(len(imgs)))
    rows = len(imgs) // cols
    w, h = imgs[0].size
    sheet = Image.new('RGB', size=(cols * w, rows * h))
    for k, img in enumerate(imgs):
        sheet.paste(img, box=(k % cols * w, k // cols * h))
    return sheet


def entity_from_row(row: Sequence[Any]) -> Optional[EntitySchema]:
    '''One table row [prompt, x, y, width, height, blend] -> EntitySchema, or None when it cannot be
    coerced (the reference reports and skips such rows, utils.py:188-196).'''
    try:
        prompt, x, y, w, h, blend = (row[k] for k in range(6))
        return EntitySchema(str(prompt).strip(), (int(x), int(y)), (int(w), int(h)), float(blend))
    except Exception as ex:          # noqa: BLE001 -- the reference catches everything here
        print('Failed to build EntitySchema:', ex)
        return None


class Runner():
    def __init__(self, state_dicts: Optional[Dict[str, dict]] = None, preset: str = 'sd15',
                 device: str = 'cuda', seed_weights: int = 0) -> None:
        if state_dicts is None:
            state_dicts = build.synthetic_state_dicts(preset, seed=seed_weights)
        self.pipe, clip, tok = build.build_models(state_dicts, preset, device)
        self.device = device
        self.encoder = CLIPEncoder(clip, tok)
        self.guide = Guide(clip, tok, device=device)
        self.generator = torch.Generator(device='cpu')     # E6: host generator
        self.eta = 0.0                                      # E5: never overwritten
        # the models and tokenizer tables are millions of long-lived objects: a full collection over them is a 50-100 ms
        # host stall between denoising loops (measured in bench.py's timed region); park them in the permanent generation
        if os.environ.get('FD_GC_FREEZE', '1') != '0':
            gc.collect()
            gc.freeze()

    def _set_seed(self, seed: Optional[int]):
        '''Falsy seed -> a random one; otherwise clamped to [0, 2^31 - 1] (utils.py:78-83).'''
        seed = min(max(seed, 0), MAX_SEED) if seed else int(torch.randint(0, MAX_SEED, (1,))[0])
        self.generator.manual_seed(seed)
        return seed

    def _run(self, batches: int, guide: GuideBase, init_image, init_size: Tuple[int, int],
             strength: float, debug: bool):
        '''`batches` sequential pipeline calls on one generator stream -- the reference's only
        data-parallel axis (utils.py:90) -- and the grid of everything they produced.'''
        images: List[Any] = []
        for _ in range(batches):
            result = self.pipe(guide=guide, init_image=init_image, init_size=init_size, strength=strength,
                               generator=self.generator, eta=self.eta, debug=debug)
            images += list(result['sample'])
        return images, image_grid(images)

    def gen(self,
            prompt='',
            init_image=None,
            guide=None,
            init_size: Tuple[int, int] = (512, 512),
            mapping_concepts: str = '',
            guide_threshold_mult: float = 0.5,
            guide_threshold_floor: float = 0.5,
            guide_clustered: float = 0.5,
            guide_linear: Tuple = (0.0, 0.5),
            guide_max_guidance: float = 0.5,
            guide_header_max: float = 0.15,
            guide_mode: int = 0,
            guide_reuse: bool = True,
            strength: float = 0.6,
            steps: int = 10,
            guidance_scale: float = 8,
            samples: int = 1,
            seed: Optional[int] = None,
            debug: bool = False):
        '''Same arguments and defaults as utils.py:114-133; returns (images, grid).'''
        given = locals()
        self._set_seed(seed)
        embeds = self.guide.embeds(prompt=prompt, guide=guide, **{k: given[k] for k in _EMBEDS_KEYS})
        return self._run(samples, SimpleGuide(self.encoder, self.pipe.unet, guidance_scale, steps, embeds),
                         init_image, init_size, strength, debug)

    def compose(self,
                bg_prompt: str = '',
                entities_df: Sequence[Sequence[Any]] = (),
                start_style: str = '',
                end_style: str = '',
                style_blend: Tuple[float, float] = (0.0, 1.0),
                init_image=None,
                batches: int = 4,
                strength: float = 0.7,
                steps: int = 30,
                guidance_scale: float = 8.0,
                init_size: Tuple[int, int] = (512, 512),
                seed: Optional[int] = None,
                debug: bool = False):
        '''Same arguments and defaults as utils.py:168-181; returns (images, grid).  `entities_df`
        rows are [prompt, offset_x, offset_y, width, height, blend]; a DataFrame is taken through
        its `_values` (utils.py:198-199); rows that do not parse or have an empty prompt are dropped.'''
        self._set_seed(seed)
        table = getattr(entities_df, '_values', entities_df)
        entities = [e for e in map(entity_from_row, table) if e is not None and e.prompt]
        self.last_schema = Schema(bg_prompt, start_style, end_style, style_blend, entities)
        guide = CompositeGuide(self.encoder, self.pipe.unet, guidance_scale, self.last_schema, steps)
        return self._run(batches, guide, init_image, init_size, strength, debug)
