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

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from . import build
from .encode.clip import CLIPEncoder
from .guidance import Guide
from .composition import CompositeGuide, EntitySchema, Schema
from .pipeline.guide import GuideBase, SimpleGuide

MAX_SEED = 2147483647


def image_grid(imgs: Sequence[Any]):
    '''Grid arrangement of PIL images: ceil(sqrt(n)) columns, n // cols rows (utils.py:36-50).'''
    from PIL import Image
    num = len(imgs)
    cols = math.ceil(num ** (1 / 2))
    rows = num // cols
    w, h = imgs[0].size
    grid = Image.new('RGB', size=(cols * w, rows * h))
    for i, img in enumerate(imgs):
        grid.paste(img, box=((i % cols) * w, (i // cols) * h))
    return grid


class Runner():
    def __init__(self, state_dicts: Optional[Dict[str, dict]] = None, preset: str = 'sd15',
                 device: str = 'cuda', seed_weights: int = 0) -> None:
        if state_dicts is None:
            state_dicts = build.synthetic_state_dicts(preset, seed=seed_weights)
        self.pipe, clip, tok = build.build_models(state_dicts, preset, device)
        self.eta = 0.0
        self.encoder = CLIPEncoder(clip, tok)
        self.guide = Guide(clip, tok, device=device)
        self.device = device
        self.generator = torch.Generator(device='cpu')

    def _set_seed(self, seed: Optional[int]):
        '''utils.py:78-83: falsy seed -> random; else clamped to [0, 2^31-1].'''
        if not seed:
            seed = int(torch.randint(0, MAX_SEED, (1,))[0])
        else:
            seed = min(max(seed, 0), MAX_SEED)
        self.generator.manual_seed(seed)
        return seed

    def _run(self, batches: int, guide: GuideBase, init_image, init_size: Tuple[int, int],
             strength: float, debug: bool):
        all_images: List[Any] = []
        for _ in range(batches):      # the reference's only data-parallel axis (utils.py:90)
            output = self.pipe(guide=guide, init_image=init_image, init_size=init_size,
                               strength=strength, generator=self.generator, eta=self.eta,
                               debug=debug)
            all_images.extend(output['sample'])
        return all_images, image_grid(all_images)

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
        self._set_seed(seed)
        guide_embeds = self.guide.embeds(
            prompt=prompt, guide=guide, mapping_concepts=mapping_concepts,
            guide_threshold_mult=guide_threshold_mult, guide_threshold_floor=guide_threshold_floor,
            guide_clustered=guide_clustered, guide_linear=guide_linear,
            guide_max_guidance=guide_max_guidance, guide_header_max=guide_header_max,
            guide_mode=guide_mode, guide_reuse=guide_reuse)
        pipeline_guide = SimpleGuide(self.encoder, self.pipe.unet, guidance_scale, steps,
                                     guide_embeds)
        return self._run(samples, pipeline_guide, init_image, init_size, strength, debug)

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
        '''Same arguments and defaults as utils.py:168-181; returns (images, grid).  Each row of
        `entities_df` is [prompt, offset_x, offset_y, width, height, blend] (a DataFrame is
        accepted through its `_values`, utils.py:198-199); a row that does not parse is reported
        and skipped, rows with an empty prompt are dropped (utils.py:188-201).'''
        self._set_seed(seed)

        def _row_to_ent(row) -> Optional[EntitySchema]:
            try:
                return EntitySchema(str(row[0]).strip(), (int(row[1]), int(row[2])),
                                    (int(row[3]), int(row[4])), float(row[5]))
            except Exception as ex:
                print('Failed to build EntitySchema:', ex)
                return None

        if hasattr(entities_df, '_values'):
            entities_df = entities_df._values
        rows = [_row_to_ent(r) for r in entities_df]
        rows = [r for r in rows if r and r.prompt]
        schema = Schema(bg_prompt, start_style, end_style, style_blend, rows)
        self.last_schema = schema
        pipeline_guide = CompositeGuide(self.encoder, self.pipe.unet, guidance_scale, schema, steps)
        return self._run(batches, pipeline_guide, init_image, init_size, strength, debug)
