'''Tokenizer stand-in with the Hugging Face call surface the reference uses
(encode/clip.py:57-63: `tokenizer(prompt, padding='max_length', max_length=...,
truncation=True, return_tensors='pt').input_ids`, `.model_max_length`).

Neither box holds the CLIP BPE vocabulary, so `SyntheticTokenizer` maps
each word to a stable pseudo-id (crc32) between the real BOS/EOS ids.  A real
`transformers.CLIPTokenizer` can be passed to every class here instead; only the call
surface above is relied on.
'''
from __future__ import annotations

import re
import zlib
from types import SimpleNamespace
from typing import List, Union

import torch

_WORD = re.compile(r"[A-Za-z]+|[0-9]|[^\sA-Za-z0-9]")


class SyntheticTokenizer():
    def __init__(self, vocab_size: int = 49408, model_max_length: int = 77):
        self.vocab_size = vocab_size
        self.model_max_length = model_max_length
        self.bos_token_id = vocab_size - 2
        self.eos_token_id = vocab_size - 1
        self.pad_token_id = vocab_size - 1

    def encode_words(self, text: str) -> List[int]:
        lo = min(1000, self.vocab_size // 8)
        span = max(1, self.vocab_size - 2 - lo)
        return [lo + zlib.crc32(w.lower().encode()) % span for w in _WORD.findall(text)]

    def __call__(self, prompt: Union[str, List[str]], padding='max_length', max_length=None,
                 truncation=True, return_tensors='pt'):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        L = max_length or self.model_max_length
        rows = []
        for p in prompts:
            ids = [self.bos_token_id] + self.encode_words(p)
            if truncation:
                ids = ids[:L - 1]
            ids = ids + [self.eos_token_id]
            ids = ids + [self.pad_token_id] * (L - len(ids))
            rows.append(ids[:L])
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
