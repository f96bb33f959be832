'''CPU oracle: PLMS (PNDM, skip_prk_steps) and K-LMS loops + CompositeGuide noise prediction
in torch fp32.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against diffusers==0.3.0 (not
installed): restated from the published algorithms; the loop plumbing follows the reference's
pipeline/flex.py:236-238,262-287 (LMS sigma scaling, t_index) and composition/guide.py:55-99.
'''
import numpy as np
import torch
from scipy import integrate

from . import ddim_ref


def pndm_loop(eps_fn, latents, steps, n_train=1000, offset=0):
    acp = ddim_ref.alphas_cumprod().numpy()
    ratio = n_train // steps
    base = np.arange(0, n_train, ratio) + offset
    ts = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()

    def prev_sample(x, t, tp, e):
        a_t, a_p = acp[t + 1 - offset], acp[tp + 1 - offset]
        cs = np.sqrt(a_p / a_t)
        den = a_t * np.sqrt(1 - a_p) + np.sqrt(a_t * (1 - a_t) * a_p)
        return float(cs) * x - float((a_p - a_t) / den) * e

    x, ets, cur, used = latents.float().clone(), [], None, []
    for counter, t in enumerate(int(v) for v in ts):
        e = eps_fn(x, t)
        used.append(t)
        prev = max(t - ratio, 0)
        if counter != 1:
            ets.append(e)
        else:
            prev, t = t, t + ratio
        if len(ets) == 1 and counter == 0:
            cur = x
        elif len(ets) == 1 and counter == 1:
            e = (e + ets[-1]) / 2
            x, cur = cur, None
        elif len(ets) == 2:
            e = (3 * ets[-1] - ets[-2]) / 2
        elif len(ets) == 3:
            e = (23 * ets[-1] - 16 * ets[-2] + 5 * ets[-3]) / 12
        else:
            e = (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4]) / 24
        x = prev_sample(x, t, prev, e)
    return x, used


def lms_loop(eps_fn, latents, steps, n_train=1000, order=4):
    acp = ddim_ref.alphas_cumprod().numpy()
    train_sigmas = ((1 - acp) / acp) ** 0.5
    ts = np.linspace(n_train - 1, 0, steps, dtype=float)
    low, high, frac = np.floor(ts).astype(int), np.ceil(ts).astype(int), np.mod(ts, 1.0)
    sigmas = np.concatenate([(1 - frac) * train_sigmas[low] + frac * train_sigmas[high], [0.0]])

    def coeff(o, t, cur):
        def f(tau):
            p = 1.0
            for k in range(o):
                if k != cur:
                    p *= (tau - sigmas[t - k]) / (sigmas[t - cur] - sigmas[t - k])
            return p
        return integrate.quad(f, sigmas[t], sigmas[t + 1], epsrel=1e-4)[0]

    x = latents.float() * float(sigmas[0])
    derivs = []
    for i, t in enumerate(ts):
        sigma = float(sigmas[i])
        e = eps_fn(x / ((sigma ** 2 + 1) ** 0.5), float(t))
        x0 = x - sigma * e
        derivs.append((x - x0) / sigma)
        if len(derivs) > order:
            derivs.pop(0)
        o = min(i + 1, order)
        x = x + sum(float(coeff(o, i, k)) * d for k, d in zip(range(o), reversed(derivs)))
    return x, sigmas


def composite_noise_pred(unet_fn, latents, uncond, bg, entities, guidance):
    '''composition/guide.py:55-99 for batch_size 1.  entities: [(embed, (ow,oh), (sw,sh), blend)].
    PINNED by tests/golden/backhalf_goldens.npz `composite/*` (the reference's own
    CompositeGuide.noise_pred with a recording stub UNet, incl. clipped and negative-offset
    boxes: torch slicing has Python's semantics, a negative start counts from the end).'''
    rows = [bg] + [e[0] for e in entities]
    cfg = guidance > 1.0
    if cfg:
        rows = [uncond] + rows
    emb = torch.cat(rows)
    out = unet_fn(torch.cat([latents] * emb.shape[0]), emb)
    stack = out[1:] if cfg else out
    bgn = stack[:1].clone()
    for k, (_, (ow, oh), (sw, sh), blend) in enumerate(entities):
        ens = stack[1 + k:2 + k, :, oh:oh + sh, ow:ow + sw]
        bgs = bgn[:, :, oh:oh + sh, ow:ow + sw]
        bgn[:, :, oh:oh + sh, ow:ow + sw] = bgs + blend * (ens - bgs)
    if cfg:
        return out[:1] + guidance * (bgn - out[:1])
    return bgn
