'''DDIM scheduler with the surface the reference pipeline uses on diffusers 0.3.0's
`DDIMScheduler` (pipeline/flex.py:55 set_format, :57-70,197 config, :177,233 set_timesteps,
:206,263 timesteps, :215 add_noise, :280-285 step(...).prev_sample).

Tables are numpy float32 like diffusers 0.3.0 (`scaled_linear` betas, cumprod); timesteps
are integers (bit-exact by construction).  `step` runs the fused HIP update kernel
(csrc/elementwise.hip k_cfg_ddim); the pipeline's fast path fuses classifier-free guidance
into the same launch.
'''
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops


class _Config(dict):
    '''Keys also read as attributes, like diffusers' FrozenDict (the reference tests
    `hasattr(scheduler.config, 'steps_offset')`, pipeline/flex.py:57).'''

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None


class _Configured():
    '''`config` reads `_internal_dict`, the attribute the reference's pipeline constructor replaces
    when it rewrites an outdated `steps_offset` (pipeline/flex.py:68-70).  A scheduler built with
    `steps_offset=0` carries NO such key -- diffusers 0.3.0, which the reference pins, has none, and
    the constructor's rewrite only fires on a key that exists.'''
    _internal_dict: _Config

    @property
    def config(self) -> _Config:
        return self._internal_dict

    def _set_config(self, steps_offset: int = 0, **kw):
        if steps_offset:
            kw['steps_offset'] = steps_offset
        self._internal_dict = _Config(**kw)


class DDIMScheduler(_Configured):
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, beta_schedule: str = 'scaled_linear',
                 clip_sample: bool = False, set_alpha_to_one: bool = False, steps_offset: int = 0,
                 prediction_type: str = 'epsilon'):
        if beta_schedule == 'scaled_linear':
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                dtype=np.float32) ** 2
        elif beta_schedule == 'linear':
            betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if clip_sample:
            raise NotImplementedError('clip_sample=True is not used by Stable Diffusion')
        self.betas = betas
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0).astype(np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        # the pinned diffusers 0.3.0 has no steps_offset and the reference calls
        # set_timesteps(steps) without one => offset 0 is the pinned behaviour
        self._set_config(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                         beta_end=beta_end, beta_schedule=beta_schedule,
                         clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                         steps_offset=steps_offset, prediction_type=prediction_type)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy()

    def set_format(self, tensor_format='pt'):
        return self

    def set_timesteps(self, num_inference_steps: int, offset: Optional[int] = None):
        '''diffusers 0.3.0: arange(0, T, T // n)[::-1] + offset.'''
        T = self.config['num_train_timesteps']
        off = self.config.get('steps_offset', 0) if offset is None else offset
        self.num_inference_steps = num_inference_steps
        self.timesteps = (np.arange(0, T, T // num_inference_steps)[::-1].copy().astype(np.int64)
                          + off)

    def _alphas(self, t: int):
        prev = t - self.config['num_train_timesteps'] // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return np.float32(a_t), np.float32(a_p)

    def step_coefficients(self, t: int, eta: float = 0.0):
        '''(c1, c2, c3, c4, sigma) fp32: x0 = (x - c1 eps)/c2 ; x' = c3 x0 + c4 eps (+ sigma z).'''
        a_t, a_p = self._alphas(int(t))
        one = np.float32(1.0)
        sigma = np.float32(0.0)
        if eta:
            var = (one - a_p) / (one - a_t) * (one - a_t / a_p)
            sigma = np.float32(eta) * np.sqrt(var, dtype=np.float32)
        return (np.sqrt(one - a_t), np.sqrt(a_t), np.sqrt(a_p),
                np.sqrt(one - a_p - sigma * sigma), sigma)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             generator=None, **_):
        c1, c2, c3, c4, sigma = self.step_coefficients(int(timestep), eta)
        B, C, H, W = sample.shape
        x = sample.to(torch.float32).clone()
        eps = model_output.to(torch.float32).contiguous()
        # NCHW eps viewed as B*C single-channel "samples" (ld = 1)
        ops.cfg_ddim_step(x, eps.view(-1, 1), B * C, 1, H * W, False, 1.0, (c1, c2, c3, c4),
                          self.config['prediction_type'] == 'v_prediction')
        if eta and float(sigma) > 0:
            gdev = getattr(generator, 'device', torch.device('cpu'))
            z = torch.randn(sample.shape, generator=generator, device=gdev).to(x.device)
            x = ops.axpby(x, z, 1.0, float(sigma))
        return SimpleNamespace(prev_sample=x)

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        t = int(timesteps.reshape(-1)[0]) if isinstance(timesteps, torch.Tensor) else int(timesteps)
        a = self.alphas_cumprod[t]
        return ops.axpby(original.to(torch.float32), noise.to(torch.float32),
                         float(np.sqrt(a)), float(np.sqrt(np.float32(1.0) - a)))


class PNDMScheduler(_Configured):
    '''PLMS branch (skip_prk_steps=True, what Stable Diffusion v1 ships and what the reference's
    `Runner` actually passes, utils.py:70) of diffusers 0.3.0's `PNDMScheduler`, restated
    from the published algorithm.  PARITY UNPINNED (diffusers is not installed); the linear
    multistep combinations run on device through fd_axpby_f32.'''
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, beta_schedule: str = 'scaled_linear',
                 skip_prk_steps: bool = True, steps_offset: int = 0):
        if beta_schedule != 'scaled_linear' or not skip_prk_steps:
            raise NotImplementedError('only the Stable-Diffusion PLMS configuration is provided')
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                            dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0).astype(np.float32)
        self._set_config(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                         beta_end=beta_end, beta_schedule=beta_schedule,
                         skip_prk_steps=skip_prk_steps, steps_offset=steps_offset)
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy()
        self.num_inference_steps = None
        self._offset = 0
        self.ets, self.counter, self.cur_sample = [], 0, None

    def set_format(self, tensor_format='pt'):
        return self

    def set_timesteps(self, num_inference_steps: int, offset: Optional[int] = None):
        T = self.config['num_train_timesteps']
        self._offset = self.config.get('steps_offset', 0) if offset is None else offset
        self.num_inference_steps = num_inference_steps
        base = np.arange(0, T, T // num_inference_steps) + self._offset
        # second timestep repeated: the first PLMS step is a two-evaluation (Heun-like) start
        self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy() \
            .astype(np.int64)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def prev_coefficients(self, t: int, t_prev: int):
        a_t = self.alphas_cumprod[t + 1 - self._offset]
        a_p = self.alphas_cumprod[t_prev + 1 - self._offset]
        one = np.float32(1.0)
        sample_coeff = np.sqrt(a_p / a_t)
        denom = a_t * np.sqrt(one - a_p) + np.sqrt(a_t * (one - a_t) * a_p)
        return np.float32(sample_coeff), np.float32(-(a_p - a_t) / denom)

    @staticmethod
    def multistep_weights(n_ets: int):
        return {2: (3 / 2, -1 / 2), 3: (23 / 12, -16 / 12, 5 / 12),
                4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[n_ets]

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, **_):
        t = int(timestep)
        ratio = self.config['num_train_timesteps'] // self.num_inference_steps
        prev = max(t - ratio, 0)
        eps = model_output.to(torch.float32).contiguous()
        sample = sample.to(torch.float32).contiguous()
        if self.counter != 1:
            self.ets.append(eps)
        else:
            prev, t = t, t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            eps = ops.axpby(eps, self.ets[-1], 0.5, 0.5)
            sample, self.cur_sample = self.cur_sample, None
        else:
            w = self.multistep_weights(min(len(self.ets), 4))
            acc = ops.axpby(self.ets[-1], self.ets[-2], w[0], w[1])
            for k in range(2, len(w)):
                acc = ops.axpby(acc, self.ets[-1 - k], 1.0, w[k])
            eps = acc
            self.ets = self.ets[-4:]
        cs, ce = self.prev_coefficients(t, prev)
        self.counter += 1
        return SimpleNamespace(prev_sample=ops.axpby(sample, eps, float(cs), float(ce)))

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        t = int(timesteps.reshape(-1)[0]) if isinstance(timesteps, torch.Tensor) else int(timesteps)
        a = self.alphas_cumprod[t]
        return ops.axpby(original.to(torch.float32), noise.to(torch.float32),
                         float(np.sqrt(a)), float(np.sqrt(np.float32(1.0) - a)))


class LMSDiscreteScheduler(_Configured):
    '''K-LMS (linear multistep, order 4) of diffusers 0.3.0, restated from the published
    algorithm.  PARITY UNPINNED.  The pipeline applies the sigma input scaling exactly where
    the reference does (pipeline/flex.py:236-238, 270-274).'''
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, beta_schedule: str = 'scaled_linear'):
        if beta_schedule != 'scaled_linear':
            raise NotImplementedError(beta_schedule)
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                            dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0).astype(np.float32)
        self.train_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.sigmas = self.train_sigmas
        self._set_config(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                         beta_end=beta_end, beta_schedule=beta_schedule)
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy()
        self.num_inference_steps = None
        self.derivatives = []

    def set_format(self, tensor_format='pt'):
        return self

    def set_timesteps(self, num_inference_steps: int):
        T = self.config['num_train_timesteps']
        self.num_inference_steps = num_inference_steps
        self.timesteps = np.linspace(T - 1, 0, num_inference_steps, dtype=float)
        low = np.floor(self.timesteps).astype(int)
        high = np.ceil(self.timesteps).astype(int)
        frac = np.mod(self.timesteps, 1.0)
        s = self.train_sigmas
        sig = (1 - frac) * s[low] + frac * s[high]
        self.sigmas = np.concatenate([sig, [0.0]])
        self.derivatives = []

    def lms_coefficient(self, order: int, t: int, current_order: int) -> float:
        from scipy import integrate

        def f(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod
        return integrate.quad(f, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, order: int = 4,
             **_):
        i = int(timestep)
        sigma = float(self.sigmas[i])
        sample = sample.to(torch.float32).contiguous()
        eps = model_output.to(torch.float32).contiguous()
        x0 = ops.axpby(sample, eps, 1.0, -sigma)
        self.derivatives.append(ops.axpby(sample, x0, 1.0 / sigma, -1.0 / sigma))
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self.lms_coefficient(order, i, k) for k in range(order)]
        out = sample
        for c, d in zip(coeffs, reversed(self.derivatives)):
            out = ops.axpby(out, d, 1.0, float(c))
        return SimpleNamespace(prev_sample=out)

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps) -> torch.Tensor:
        i = int(timesteps.reshape(-1)[0]) if isinstance(timesteps, torch.Tensor) else int(timesteps)
        return ops.axpby(original.to(torch.float32), noise.to(torch.float32), 1.0,
                         float(self.sigmas[i]))
