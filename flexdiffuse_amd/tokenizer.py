'''Tokenizer stand-in with the Hugging Face call surface the reference uses
(encode/clip.py:57-63: `tokenizer(prompt, padding='max_length', max_length=...,
truncation=True, return_tensors='pt').input_ids`, `.model_max_length`).

Neither box holds the CLIP BPE vocabulary, so `SyntheticTokenizer` maps
each word to a stable pseudo-id (crc32) between the real BOS/EOS ids.  Where the vocabulary
IS on disk (vocab.json + merges.txt of "openai/clip-vit-large-patch14", what the reference's
`Runner.__init__` downloads, utils.py:24-25, 61-68), `CLIPBPETokenizer` is the byte-level BPE
itself: no transformers import on the product path.  It is checked id-for-id against
`transformers.CLIPTokenizer` on a toy vocabulary in tests/test_tokenizer.py.  A real
`transformers.CLIPTokenizer` can also be passed to every class here; only the call surface
above is relied on.
'''
from __future__ import annotations

import json
import os
import re
import unicodedata
import zlib
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

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


def _byte_alphabet() -> Dict[int, str]:
    '''The byte -> printable-character table of byte-level BPE (GPT-2 / CLIP): the printable
    Latin-1 bytes stand for themselves, the other 68 are moved to U+0100 onwards in byte order.'''
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, spare = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + spare)
            spare += 1
    return table


class CLIPBPETokenizer():
    '''CLIP's lower-cased byte-level BPE with the Hugging Face call surface.

    Text -> NFC, runs of whitespace -> one space, lower case -> pieces (the two special tokens,
    the English clitics 's 't 're 've 'm 'll 'd, letter runs, SINGLE digits, runs of other
    non-space characters) -> UTF-8 bytes as printable characters, the last one of each piece
    carrying "</w>" -> greedy lowest-rank pair merges -> vocabulary ids.  Rows are
    `[BOS] ids [EOS]`, truncated so that EOS survives, padded with `pad_token` (the EOS token
    for SD1.x; "!" = id 0 for the SD2.x tokenizer).

    Which Hugging Face path this mirrors.  `text_cleanup='fast'` (default of this class) is the pipeline of
    the `tokenizers`-backed CLIPTokenizer (`CLIPTokenizerFast`; the only CLIPTokenizer of
    transformers >= 5, and what the slow class does when `ftfy` is installed): normalizers NFC -> `\\s+` -> " "
    -> lower case, then the piece pattern -- checked id-for-id against the installed transformers in
    tests/test_tokenizer.py, CJK and control characters included.
    `text_cleanup='basic'` is the mode that matches the REFERENCE'S PINNED STACK: transformers 4.21.1 calls the
    SLOW `CLIPTokenizer` (utils.py:61-63) and `ftfy` is not in its requirements, so the text first goes through
    BERT's BasicTokenizer -- the full algorithm, restated here from the published source: control / NUL / U+FFFD
    characters dropped, spaces around every CJK ideograph, lower-casing, NFD + removal of combining marks
    (ACCENTS STRIPPED: "café" -> "cafe"), and EVERY punctuation character split off as its own token
    ("it's" -> it ' s, so the clitic pieces of the pattern never fire).  The two modes therefore DIFFER on prompts
    with apostrophes, accents or punctuation inside words; they agree on plain lower-ASCII words and spaces.
    `build.from_directories` / `Runner` / `FlexPipeline.from_pretrained` default to 'basic' for that reason.
    (transformers 4.21.1 is not installed, so only known answers pin the 'basic' mode.)  Needs the third-party
    `regex` module (Unicode property classes in the piece pattern), as transformers' own slow tokenizer does.'''

    BOS, EOS = '<|startoftext|>', '<|endoftext|>'

    def __init__(self, vocab: Dict[str, int], merges: Iterable[Union[str, Sequence[str]]],
                 model_max_length: int = 77, pad_token: Optional[str] = None, text_cleanup: str = 'fast'):
        try:
            import regex
        except ImportError as ex:      # pragma: no cover
            raise ImportError('CLIPBPETokenizer needs the `regex` package (\\p{L} / \\p{N} classes in '
                              "CLIP's piece pattern)") from ex
        if text_cleanup not in ('fast', 'basic'):
            raise ValueError("text_cleanup must be 'fast' or 'basic'")
        self.text_cleanup = text_cleanup
        self.vocab = dict(vocab)
        pairs = []
        for m in merges:
            if isinstance(m, str):
                if not m.strip() or m.startswith('#version'):
                    continue
                m = m.split()
            pairs.append((m[0], m[1]))
        self.rank: Dict[Tuple[str, str], int] = {p: i for i, p in enumerate(pairs)}
        self.model_max_length = model_max_length
        self.vocab_size = len(self.vocab)
        self.bos_token_id = self.vocab[self.BOS]
        self.eos_token_id = self.vocab[self.EOS]
        self.unk_token_id = self.eos_token_id
        self.pad_token_id = self.vocab[pad_token] if pad_token is not None else self.eos_token_id
        self._bytes = _byte_alphabet()
        self._pieces = regex.compile(
            r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|\p{L}+|\p{N}|[^\s\p{L}\p{N}]+")
        self._space = regex.compile(r'\s+')
        self._specials = regex.compile(r'(<\|startoftext\|>|<\|endoftext\|>)')
        self._memo: Dict[str, List[str]] = {}

    @classmethod
    def from_pretrained(cls, directory: str, **kw) -> 'CLIPBPETokenizer':
        '''`directory` holds vocab.json and merges.txt (the layout of the tokenizer/ folder of a
        diffusers checkpoint, or of the CLIP model repository).'''
        with open(os.path.join(directory, 'vocab.json'), encoding='utf-8') as f:
            vocab = json.load(f)
        with open(os.path.join(directory, 'merges.txt'), encoding='utf-8') as f:
            merges = f.read().split('\n')
        cfg = os.path.join(directory, 'tokenizer_config.json')
        if os.path.exists(cfg) and 'pad_token' not in kw:
            with open(cfg, encoding='utf-8') as f:
                pad = json.load(f).get('pad_token')
            if isinstance(pad, dict):
                pad = pad.get('content')
            if isinstance(pad, str) and pad in vocab:
                kw['pad_token'] = pad
        return cls(vocab, merges, **kw)

    def _merge(self, piece: str) -> List[str]:
        got = self._memo.get(piece)
        if got is not None:
            return got
        sym = [self._bytes[b] for b in piece.encode('utf-8')]
        sym[-1] += '</w>'
        while len(sym) > 1:
            best, at = None, -1
            for i in range(len(sym) - 1):
                r = self.rank.get((sym[i], sym[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            a, b = sym[at], sym[at + 1]
            out, i = [], 0
            while i < len(sym):                      # merge EVERY occurrence of the winning pair
                if i + 1 < len(sym) and sym[i] == a and sym[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(sym[i])
                    i += 1
            sym = out
        self._memo[piece] = sym
        return sym

    @staticmethod
    def _is_cjk(cp: int) -> bool:
        return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or
                0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or
                0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)

    @classmethod
    def _basic_clean(cls, text: str) -> str:
        '''BasicTokenizer's character clean-ups: drop NUL / U+FFFD / control characters (category C*,
        except tab / newline / carriage return, which count as whitespace), map whitespace to a space,
        and put spaces around CJK ideographs.'''
        out = []
        for ch in text:
            cp = ord(ch)
            if ch in '\t\n\r' or ch == ' ' or unicodedata.category(ch) == 'Zs':
                out.append(' ')
            elif cp == 0 or cp == 0xFFFD or unicodedata.category(ch).startswith('C'):
                continue
            elif cls._is_cjk(cp):
                out.append(' ' + ch + ' ')
            else:
                out.append(ch)
        return ''.join(out)

    @staticmethod
    def _is_punctuation(ch: str) -> bool:
        cp = ord(ch)
        if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
            return True                  # every non-alphanumeric ASCII symbol counts ("$", "^", "`" too)
        return unicodedata.category(ch).startswith('P')

    @classmethod
    def _basic_tokens(cls, text: str) -> List[str]:
        '''BERT BasicTokenizer(do_lower_case=True) as published: character clean-ups, whitespace split,
        then per token lower case, accent stripping (NFD, combining marks dropped) and a split at EVERY
        punctuation character.  "don't" -> don ' t;  "café" -> cafe.'''
        out: List[str] = []
        for tok in cls._basic_clean(text).split():
            tok = ''.join(c for c in unicodedata.normalize('NFD', tok.lower())
                          if unicodedata.category(c) != 'Mn')
            word = ''
            for c in tok:
                if cls._is_punctuation(c):
                    if word:
                        out.append(word)
                    out.append(c)
                    word = ''
                else:
                    word += c
            if word:
                out.append(word)
        return out

    def tokenize(self, text: str) -> List[str]:
        if self.text_cleanup == 'basic':
            # the slow tokenizer cuts the registered special tokens out of the text first and hands only
            # the stretches between them to BasicTokenizer; no NFC afterwards
            parts = self._specials.split(text)
            text = ' '.join(p if p in (self.BOS, self.EOS) else ' '.join(self._basic_tokens(p)) for p in parts)
            text = self._space.sub(' ', text).strip()
        else:
            text = self._space.sub(' ', unicodedata.normalize('NFC', text)).lower()
        out: List[str] = []
        for piece in self._pieces.findall(text):
            if piece in (self.BOS, self.EOS):
                out.append(piece)
            else:
                out.extend(self._merge(piece))
        return out

    def encode(self, text: str) -> List[int]:
        return [self.vocab.get(t, self.unk_token_id) for t in self.tokenize(text)]

    def __call__(self, prompt: Union[str, List[str]], padding='max_length', max_length=None,
                 truncation=True, return_tensors='pt'):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        L = max_length or self.model_max_length
        rows = []
        for p in prompts:
            ids = self.encode(p)
            if truncation:
                ids = ids[:L - 2]
            rows.append([self.bos_token_id] + ids + [self.eos_token_id])
        width = L if padding == 'max_length' else max(len(r) for r in rows)
        rows = [r + [self.pad_token_id] * (width - len(r)) for r in rows]
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
