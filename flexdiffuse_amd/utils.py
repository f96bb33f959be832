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
    def __init__(self, local: bool = True, device: str = 'cuda', *, sd_dir: Optional[str] = None,
                 clip_dir: Optional[str] = None, tokenizer_dir: Optional[str] = None,
                 state_dicts: Optional[Dict[str, dict]] = None, preset: str = 'sd15', seed_weights: int = 0,
                 freeze_gc: bool = False, pause_gc: bool = False, scheduler=None, text_cleanup: str = 'basic') -> None:
        '''`Runner(local=True, device='cuda')` as the reference constructs it (utils.py:54-76); the hub
        ids the reference hard-codes (utils.py:24-25) become local directories:
          sd_dir        diffusers-layout checkpoint (unet/, vae/, tokenizer/) -- env FD_SD_DIR
          clip_dir      CLIPModel directory                                    -- env FD_CLIP_DIR
          tokenizer_dir vocab.json + merges.txt; default `sd_dir`/tokenizer     -- env FD_TOKENIZER_DIR
        The scheduler is the checkpoint's own (`sd_dir`/scheduler/scheduler_config.json: PNDM/PLMS for SD-v1-4, as the
        reference's `sd.scheduler`, utils.py:70) unless `scheduler=` overrides it; `text_cleanup='basic'` is the
        tokenization of the reference's pinned slow CLIPTokenizer ('fast': CLIPTokenizerFast).
        `local=False` (the reference's --dl) cannot be served: there is no hub access here.
        Without directories: `state_dicts` (fp32 CPU tensors with HF key names) or seeded synthetic
        weights of `preset` with the synthetic tokenizer.
        `freeze_gc` / `pause_gc` (both off by default: they change interpreter-global state): park the
        long-lived model objects in the GC's permanent generation / keep the cyclic GC off across each
        denoising loop -- what `bench.py` runs with.'''
        if not local:
            raise RuntimeError('Runner(local=False): no network access -- place the checkpoint on disk and pass '
                               'sd_dir / clip_dir (or set FD_SD_DIR / FD_CLIP_DIR)')
        sd_dir = sd_dir or os.environ.get('FD_SD_DIR')
        clip_dir = clip_dir or os.environ.get('FD_CLIP_DIR')
        tokenizer_dir = tokenizer_dir or os.environ.get('FD_TOKENIZER_DIR')
        if sd_dir or clip_dir:
            if not (sd_dir and clip_dir):
                raise ValueError('Runner needs both sd_dir and clip_dir (utils.py:24-25: two checkpoints)')
            self.pipe, clip, tok = build.from_directories(sd_dir, clip_dir, tokenizer_dir, preset=preset,
                                                          device=device, scheduler=scheduler, text_cleanup=text_cleanup)
        else:
            if state_dicts is None:
                state_dicts = build.synthetic_state_dicts(preset, seed=seed_weights)
            tok = build.load_tokenizer(tokenizer_dir, text_cleanup) if tokenizer_dir else None
            self.pipe, clip, tok = build.build_models(state_dicts, preset, device, tokenizer=tok, scheduler=scheduler)
        self.pipe.pause_gc = pause_gc
        self.device = device
        self.encoder = CLIPEncoder(clip, self.pipe.tokenizer)     # utils.py:73-74
        self.guide = Guide(clip, self.pipe.tokenizer, device=device)
        self.generator = torch.Generator(device='cpu')     # E6: host generator
        self.eta = 0.0                                      # E5: never overwritten
        # the models and tokenizer tables are millions of long-lived objects: a full collection over them is a 50-100 ms
        # host stall between denoising loops (measured in bench.py's timed region)
        if freeze_gc:
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
