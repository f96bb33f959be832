'''CPU oracle: guidance map + tween (front half of the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, in this project's
own words, the algorithm of the reference file `guidance.py`:

  map_emb             <- guidance.py:23-85   (_map_emb)
  clustered_weights   <- guidance.py:88-172  (_traverse_a_to_b, _clustered_guidance)
  blend_weights       <- guidance.py:175-193 (_blend_weights)
  tween               <- guidance.py:196-272 (Tweener.tween)
  concept_override    <- guidance.py:275-312 (ConceptMapper)

PINNED: every function here is checked against golden vectors captured from the
reference itself (tests/golden/guidance_*.npz, tests/test_oracle_guidance.py).

Arithmetic follows the reference: similarities in fp32 (one mat-vec per guide
token, then a 77-way softmax), decisions on the float64 images of those fp32
values, weights in fp32, final blend in fp32 without fused multiply-add.
'''
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

ORDER_TEXT = 0
ORDER_ALIGN = 1
ORDER_DIRECT = 2


def _as_2d(x) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    t = t.detach().to('cpu', torch.float32)
    if t.dim() == 3:
        assert t.shape[0] == 1, 'oracle maps one prompt at a time'
        t = t[0]
    return t


def similarity(alt, txt) -> np.ndarray:
    '''S'[i, j] = softmax_j'(100 * cos(alt_i, txt_j'))[j + 1]  -> (N, L-1) float64.

    guidance.py:43-56: rows are L2-normalised in fp32, one mat-vec + softmax per
    guide token, and the header column (text token 0) is dropped.
    '''
    a = _as_2d(alt)
    t = _as_2d(txt)
    a = a / a.norm(dim=-1, keepdim=True)
    t = t / t.norm(dim=-1, keepdim=True)
    tt = t.mT.contiguous()
    rows = []
    for i in range(a.shape[0]):
        logits = 100.0 * (a[i:i + 1] @ tt)
        rows.append(logits.softmax(dim=-1)[0, 1:])
    return torch.stack(rows).numpy().astype(np.float64)


def assign(sim: np.ndarray, n_text: int, reuse: bool, order: int) -> np.ndarray:
    '''Greedy assignment over the (N, L-1) similarity table -> (L, 2) float64.

    guidance.py:57-85.  Row j of the result describes S' column j (i.e. text token
    j+1, the reference's off-by-one); the last row is never written.
    A slot only "locks" once it holds s > 0, and (without reuse) every accepted
    candidate consumes its guide token even if it did not lock the slot.
    '''
    n_alt, n_col = sim.shape
    out = np.zeros((n_text, 2), dtype=np.float64)
    if order == ORDER_DIRECT:
        for j in range(min(n_alt, n_col)):
            out[j] = (j, sim[j, j])
        return out
    ii, jj = np.meshgrid(np.arange(n_alt), np.arange(n_col), indexing='ij')
    ii, jj, ss = ii.ravel(), jj.ravel(), sim.ravel()
    if order == ORDER_TEXT:
        perm = np.lexsort((ii, -ss, jj))      # text asc, s desc, guide asc
    else:
        perm = np.lexsort((ii, jj, -ss))      # s desc, text asc, guide asc
    used = np.zeros(n_alt, dtype=bool)
    for k in perm:
        i, j = ii[k], jj[k]
        if out[j, 1] > 0 or used[i]:
            continue
        out[j] = (i, ss[k])
        if not reuse:
            used[i] = True
    return out


def map_emb(alt, txt, reuse: bool = True, order: int = ORDER_ALIGN) -> np.ndarray:
    '''guidance.py:23-85 -> (L, 2) float64 rows of (guide index, similarity).'''
    n_text = _as_2d(txt).shape[0]
    return assign(similarity(alt, txt), n_text, reuse, order)


def clustered_weights(mapped: np.ndarray, threshold: float,
                      gain: float) -> Optional[torch.Tensor]:
    '''guidance.py:88-172: 1 at similarity peaks, sliding linearly to 0 at the
    valleys between them, times `gain`; None when there is no peak.  Raises
    ZeroDivisionError for adjacent equal peaks exactly like the reference.'''
    n = mapped.shape[0]
    s = mapped[:, 1]
    peaks = [k for k in range(1, n - 1)
             if not (s[k] < threshold) and s[k - 1] <= s[k] >= s[k + 1]]
    if not peaks:
        return None
    valleys = []
    if peaks[0] != 0:
        valleys.append(0)
    for p, q in zip(peaks[:-1], peaks[1:]):
        if q - p > 0:
            valleys.append(p + math.ceil((q - p) / 2))
    if peaks[-1] != n - 1:
        valleys.append(n - 1)
    w = torch.ones((n,))
    if valleys[0] == 0:
        w[0] -= 1.0
    vi = 0
    for p in peaks:
        v = valleys[vi]
        if v < p:                                # slide down to the left valley
            g = 1.0 / (p - v)
            for k in range(1, p - v):
                w[p - k] -= g * k
            vi += 1
        if vi >= len(valleys):
            break
        v = valleys[vi]                          # slide down to the right valley
        g = 1.0 / (v - p)
        for k in range(1, v - p + 1):
            w[p + k] -= g * k
    return w * gain


def blend_weights(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    '''guidance.py:175-193.'''
    assert a.shape == b.shape
    if a.max() >= 0:
        return torch.maximum(a, b) if b.max() >= 0 else a + b
    return torch.minimum(a, b)


def tween_weights(mapped: np.ndarray,
                  threshold: Tuple[float, float] = (0.5, 0.5),
                  linear: Tuple[float, float] = (0.0, 0.5),
                  clustered: float = 0.5,
                  header_max: float = 0.15,
                  linear_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    '''guidance.py:219-254 -> fp32 (L,) blend weights before the max_guidance cap.'''
    floor, mult = threshold
    n = mapped.shape[0]
    w = (torch.linspace(linear[0], linear[1], steps=n)
         if linear_weights is None else linear_weights.clone().float())
    if clustered != 0:
        cw = clustered_weights(mapped, mapped[:, 1].mean(), clustered)
        if cw is not None:
            w = blend_weights(w, cw)
    if mult != 0:
        th = torch.ones_like(w) * mult
        th[torch.from_numpy(mapped[:, 1] < floor)] = 0
        w = blend_weights(w, th)
    if header_max < 1.0:
        h = w[0].item()
        w[0] = min(h, header_max) if h >= 0 else max(h, -header_max)
    return w


def tween(base, alt,
          threshold: Tuple[float, float] = (0.5, 0.5),
          linear: Tuple[float, float] = (0.0, 0.5),
          clustered: float = 0.5,
          max_guidance: float = 0.5,
          header_max: float = 0.15,
          order: int = ORDER_ALIGN,
          reuse: bool = True,
          linear_weights: Optional[torch.Tensor] = None,
          mapped: Optional[np.ndarray] = None):
    '''guidance.py:215-272.  Returns (out (1,L,D) fp32, weights (L,), mapped (L,2)).'''
    b = _as_2d(base)
    a = _as_2d(alt)
    if mapped is None:
        mapped = map_emb(a, b, reuse, order)
    w = tween_weights(mapped, threshold, linear, clustered, header_max, linear_weights)
    out = torch.zeros_like(b)
    for j in range(b.shape[0]):
        i, s = int(mapped[j, 0]), mapped[j, 1]
        iw = min(w[j].item(), max_guidance)
        if iw == 0:
            out[j] = b[j]
        elif abs(iw) >= 1.0 - s:
            out[j] = a[i]
        else:
            out[j] = b[j] + (a[i] - b[j]) * iw
    return out[None], w, mapped


def concept_override(guide, concept, base, out=None, verbose=False) -> torch.Tensor:
    '''guidance.py:275-312 (ConceptMapper.__init__ + .map): where a text token
    aligns > 0.9 with a concept token, replace it by that concept's guide token.'''
    g = _as_2d(guide)
    c = _as_2d(concept)
    b = _as_2d(base)
    res = (b.clone() if out is None else _as_2d(out).clone())
    cm = map_emb(g, c, False, ORDER_TEXT)
    ct = map_emb(c, b, True, ORDER_ALIGN)
    for j in range(ct.shape[0]):
        ci = int(ct[j, 0])
        if ci - 1 < 0:
            continue
        if ct[j, 1] > 0.9:
            res[j + 1] = g[int(cm[ci - 1, 0])]
    return res[None]
