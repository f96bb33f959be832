'''Data contract of region-composited guidance.

Field names, order and defaults are those of the reference's composition/schema.py:6-26
(`Runner.compose` builds these from table rows, utils.py:188-201, and `CompositeGuide`
reads them); positions and sizes are in IMAGE pixels and are floor-divided by 8 into latent
blocks by the guide.
'''
from __future__ import annotations

import dataclasses
import json
from typing import List, Tuple

Pixels = Tuple[int, int]


@dataclasses.dataclass
class EntitySchema():
    '''One prompt painted into a rectangle of the canvas.'''
    prompt: str
    offset: Pixels               # (x, y) of the box's top-left corner
    size: Pixels                 # (width, height) of the box
    blend: float = 0.8           # 0 = background only ... 1 = entity only, inside the box

    def __post_init__(self):
        if len(self.offset) != 2 or len(self.size) != 2:
            raise ValueError('offset and size are (x, y) / (width, height) pairs')


@dataclasses.dataclass
class Schema():
    '''A background prompt, two style prompts with their blend range, and the entities.'''
    background_prompt: str
    style_start_prompt: str
    style_end_prompt: str
    style_blend: Tuple[float, float]
    entities: List[EntitySchema]

    def json(self) -> str:
        '''Same JSON shape as the reference's `Schema.json()` (entities as plain dicts).'''
        return json.dumps(dataclasses.asdict(self))
