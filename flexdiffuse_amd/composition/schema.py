'''Composition schema -- mirror of the reference's composition/schema.py:6-26.'''
import json
from dataclasses import dataclass
from typing import List, Tuple


@dataclass
class EntitySchema():
    prompt: str
    offset: Tuple[int, int]      # (x, y) pixels
    size: Tuple[int, int]        # (w, h) pixels
    blend: float = 0.8


@dataclass
class Schema():
    background_prompt: str
    style_start_prompt: str
    style_end_prompt: str
    style_blend: Tuple[float, float]
    entities: List[EntitySchema]

    def json(self) -> str:
        d = dict(self.__dict__)
        d['entities'] = [e.__dict__ for e in self.entities]
        return json.dumps(d)
