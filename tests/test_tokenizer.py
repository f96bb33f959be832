'''CLIPBPETokenizer against transformers.CLIPTokenizer (the library the reference calls,
encode/clip.py:57-63; transformers is the third-party oracle here, test-only) on a toy
vocabulary: byte alphabet + merges learned from a small corpus by a plain BPE trainer.'''
import collections
import json
import random

import pytest
import torch

from flexdiffuse_amd.tokenizer import CLIPBPETokenizer, _byte_alphabet

CORPUS = ('a photo of a cat sitting on the table . an oil painting of the mountains at sunset , '
          'highly detailed , trending on artstation . a dog\'s portrait , it\'s what we\'ve seen ; '
          'the painter\'ll paint 1920s cars & 42 bicycles ! café naïve über straße '
          '日本語 русский photo photograph photographic '
          'painting painted painter mountains mountain table tables sunset sunrise').split()


def toy_vocab(n_merges=120):
    alpha = list(_byte_alphabet().values())
    vocab = {}
    for ch in alpha:
        vocab[ch] = len(vocab)
    for ch in alpha:
        vocab[ch + '</w>'] = len(vocab)
    words = collections.Counter()
    table = _byte_alphabet()
    for w in CORPUS:
        sym = [table[b] for b in w.lower().encode('utf-8')]
        sym[-1] += '</w>'
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for sym, c in words.items():
            for a, b in zip(sym, sym[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _c = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        merges.append((a, b))
        if a + b not in vocab:
            vocab[a + b] = len(vocab)
        new = collections.Counter()
        for sym, c in words.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and sym[i] == a and sym[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(sym[i]); i += 1
            new[tuple(out)] += c
        words = new
    vocab['<|startoftext|>'] = len(vocab)
    vocab['<|endoftext|>'] = len(vocab)
    return vocab, merges


PROMPTS = [
    'a photo of a cat', 'A  Photo\tof\nthe   TABLE!!', "it's the painter's dog, we've seen it; I'll paint",
    'cars from the 1920s & 42 bicycles', 'café naïve über straße',
    '日本語 and русский', '', '   ', '{}',
    'photo<|endoftext|>graph', 'x' * 300, 'a ' * 100, 'mountains... at sunset?! (highly-detailed)', "don't 'quote' 'd",
    '\U0001F600 emoji  nbsp', '3.14159 2+2=4 #tag @user',
]


@pytest.fixture(scope='module')
def pair():
    transformers = pytest.importorskip('transformers')
    vocab, merges = toy_vocab()
    mine = CLIPBPETokenizer(vocab, merges)
    try:
        ref = transformers.CLIPTokenizer(vocab=vocab, merges=merges)
    except Exception:
        ref = transformers.CLIPTokenizer(vocab=vocab, merges=[' '.join(m) for m in merges])
    return mine, ref


def test_ids_equal_transformers(pair):
    mine, ref = pair
    assert mine.bos_token_id == ref.bos_token_id and mine.eos_token_id == ref.eos_token_id
    assert mine.pad_token_id == ref.pad_token_id
    for p in PROMPTS:
        a = mine(p, padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
        b = ref(p, padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
        assert a.shape == (1, 77) and a.dtype == torch.long
        assert torch.equal(a, b), (p, a[0, :20].tolist(), b[0, :20].tolist())


def test_batch_and_random_strings(pair):
    mine, ref = pair
    rng = random.Random(3)
    words = CORPUS + ['zzz', 'Qq', "o'clock", '77', 'x-y', '...']
    prompts = [' '.join(rng.choice(words) for _ in range(rng.randint(1, 90))) for _ in range(40)]
    a = mine(prompts, padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
    b = ref(prompts, padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
    assert torch.equal(a, b)
    # EOS survives truncation: a full row ends in EOS, and BOS leads
    assert (a[:, 0] == mine.bos_token_id).all() and (a[:, -1] == mine.eos_token_id).all()


def test_from_pretrained_layout(tmp_path):
    vocab, merges = toy_vocab(40)
    (tmp_path / 'vocab.json').write_text(json.dumps(vocab), encoding='utf-8')
    (tmp_path / 'merges.txt').write_text('#version: 0.2\n' + '\n'.join(' '.join(m) for m in merges) + '\n',
                                         encoding='utf-8')
    (tmp_path / 'tokenizer_config.json').write_text(json.dumps({'pad_token': '!'}))
    from flexdiffuse_amd import build
    tok = build.load_tokenizer(str(tmp_path))
    assert tok.model_max_length == 77 and tok.pad_token_id == vocab['!']   # the SD2.x padding token
    ids = tok('a photo', padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
    direct = CLIPBPETokenizer(vocab, merges, pad_token='!')('a photo').input_ids
    assert torch.equal(ids, direct) and ids[0, -1] == vocab['!']
    with pytest.raises(FileNotFoundError):
        build.load_tokenizer(str(tmp_path / 'missing'))


def test_basic_cleanup_mode_known_answers():
    '''text_cleanup='basic': the slow CLIPTokenizer's no-ftfy path (BERT BasicTokenizer first) that the
    reference's pinned transformers 4.21.1 runs -- control / NUL / U+FFFD characters dropped, every CJK
    ideograph its own piece -- against known answers of the published algorithm; accents stripped, every
    punctuation character its own piece; identical to the default mode on plain alphanumeric prompts.'''
    vocab, merges = toy_vocab()
    fast, basic = CLIPBPETokenizer(vocab, merges), CLIPBPETokenizer(vocab, merges, text_cleanup='basic')
    assert CLIPBPETokenizer._basic_clean('a\x00b\x07c\ufffdd') == 'abcd'
    assert CLIPBPETokenizer._basic_clean('x\ty\u00a0z') == 'x y z'                 # tab and NBSP (Zs) are whitespace
    assert CLIPBPETokenizer._basic_clean('tab\u200bzero') == 'tabzero'             # U+200B is category Cf
    assert CLIPBPETokenizer._basic_clean('日本語x') == ' 日  本  語 x'
    assert CLIPBPETokenizer._basic_clean('한국어') == '한국어'                          # Hangul is not in the CJK ranges
    # one piece per ideograph instead of one for the run
    assert basic.tokenize('日本語') == [t for ch in '日本語' for t in fast.tokenize(ch)]
    assert fast.tokenize('日本語') != basic.tokenize('日本語')
    assert basic.encode('a\x00b') == fast.encode('ab') != fast.encode('a\x00b')
    # BasicTokenizer(do_lower_case=True) also strips accents and splits at every punctuation character
    assert CLIPBPETokenizer._basic_tokens("Don't stop") == ['don', "'", 't', 'stop']
    assert CLIPBPETokenizer._basic_tokens('Café naïve, über!') == ['cafe', 'naive', ',', 'uber', '!']
    assert CLIPBPETokenizer._basic_tokens('wait... what?!') == ['wait', '.', '.', '.', 'what', '?', '!']
    assert CLIPBPETokenizer._basic_tokens('a$b^c`d') == ['a', '$', 'b', '^', 'c', '`', 'd']
    assert basic.encode("it's") == fast.encode('it') + fast.encode("'") + fast.encode('s') != fast.encode("it's")
    assert basic.encode('café') == fast.encode('cafe') != fast.encode('café')
    assert basic.encode('mountains... at') == fast.encode('mountains') + 3 * fast.encode('.') + fast.encode('at')
    # registered special tokens are cut out before BasicTokenizer sees the text
    assert basic.encode('photo<|endoftext|>graph') == fast.encode('photo<|endoftext|>graph')
    import unicodedata
    plain = lambda p: all(c == ' ' or (c.isascii() and c.isalnum()) for c in p)
    for p in PROMPTS + ['a photo of a cat sitting on the table', 'oil painting of 42 mountains']:
        if plain(p):
            assert basic.encode(p) == fast.encode(p), p
    with pytest.raises(ValueError):
        CLIPBPETokenizer(vocab, merges, text_cleanup='ftfy')
