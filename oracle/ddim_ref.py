'''CPU oracle: DDIM scheduler arithmetic (numpy/torch fp32).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against diffusers==0.3.0 `DDIMScheduler`
(not installed); formula restated from the DDIM paper / SURVEY.md App. C and anchored
by analytic known answers there (alphas_cumprod values, timestep tables, eps=0
invariant).  Call sites in the reference: pipeline/flex.py:177,206,215,280-285.
'''
import numpy as np
import torch


def alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    '''scaled_linear betas; diffusers 0.3.0 schedulers are numpy-float32 based
    (`set_format('pt')`, pipeline/flex.py:55), so the table is built with numpy.'''
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float32) ** 2
    return torch.from_numpy(np.cumprod(1.0 - betas, axis=0).astype(np.float32))


def timesteps(num_inference_steps: int, n_train=1000, steps_offset=0) -> np.ndarray:
    '''diffusers 0.3.0 set_timesteps: arange(0, T, T // n)[::-1] + offset (n=30 therefore
    yields 31 timesteps).'''
    return np.arange(0, n_train, n_train // num_inference_steps)[::-1].copy().astype(np.int64) \
        + steps_offset


def ddim_step(eps, t: int, x, acp, num_inference_steps, n_train=1000, set_alpha_to_one=False,
              prediction_type='epsilon', eta: float = 0.0, noise=None):
    '''DDIM update x_t -> x_{t-1} (DDIM paper eq. 12).  eta = 0: deterministic.  eta > 0 (the reference passes
    it through, pipeline/flex.py:247-251,280-285): sigma_t = eta sqrt((1 - a_prev) / (1 - a_t)) sqrt(1 - a_t / a_prev),
    the direction term shrinks to sqrt(1 - a_prev - sigma_t^2) eps and sigma_t * `noise` is added.'''
    prev = t - n_train // num_inference_steps
    a_t = acp[t]
    a_prev = acp[prev] if prev >= 0 else (torch.tensor(1.0) if set_alpha_to_one else acp[0])
    if prediction_type == 'v_prediction':
        x0 = a_t.sqrt() * x - (1 - a_t).sqrt() * eps
        eps = a_t.sqrt() * eps + (1 - a_t).sqrt() * x
    else:
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
    if not eta:
        return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps
    sigma = eta * (((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)).sqrt()
    return a_prev.sqrt() * x0 + (1 - a_prev - sigma ** 2).sqrt() * eps + sigma * noise


def add_noise(x, noise, t, acp):
    a = acp[t]
    return a.sqrt() * x + (1 - a).sqrt() * noise
