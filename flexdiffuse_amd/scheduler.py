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
    __getattr__ = dict.__getitem__


class DDIMScheduler():
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
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
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
        off = self.config['steps_offset'] if offset is None else offset
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
