'''Analytic known answers for the DDIM restatement.  CPU only.'''
import numpy as np
import torch

from oracle import ddim_ref as D


def test_alphas_cumprod_kats():
    acp = D.alphas_cumprod()
    kats = {0: 0.99914998, 1: 0.99829602, 20: 0.98131436, 500: 0.27633256, 980: 0.00584377,
            981: 0.00577549, 999: 0.00466009}
    for t, v in kats.items():
        assert abs(acp[t].item() - v) < 5e-8, (t, acp[t].item())


def test_timestep_tables():
    t50 = D.timesteps(50)
    assert t50[0] == 980 and t50[-1] == 0 and len(t50) == 50 and (np.diff(t50) == -20).all()
    t50o = D.timesteps(50, steps_offset=1)
    assert t50o[0] == 981 and t50o[-1] == 1
    assert list(D.timesteps(10)) == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    t30 = D.timesteps(30)
    assert len(t30) == 31 and t30[0] == 990 and t30[1] == 957 and t30[-1] == 0


def test_step_invariants():
    acp = D.alphas_cumprod()
    x = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    out = D.ddim_step(torch.zeros_like(x), 500, x, acp, 50)
    assert torch.allclose(out, (acp[480] / acp[500]).sqrt() * x, rtol=1e-6, atol=1e-7)
    last = D.ddim_step(torch.zeros_like(x), 0, x, acp, 50)
    assert torch.allclose(last, x, rtol=1e-6)      # prev<0 -> alpha_prev = acp[0]
    n = torch.randn_like(x)
    noisy = D.add_noise(x, n, 580, acp)
    assert torch.allclose(noisy, acp[580].sqrt() * x + (1 - acp[580]).sqrt() * n)


def test_eta_step_known_answers():
    '''eta > 0 (DDIM paper eq. 12 / 16; the reference forwards eta, pipeline/flex.py:247-251): eta = 0 and
    noise = None is the deterministic update; for eta = 1 sigma_t^2 is the DDPM posterior variance
    (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev); the marginal variance of x_{t-1} given x0 stays 1 - a_prev;
    and the product scheduler's host coefficients are the same numbers.'''
    from flexdiffuse_amd.scheduler import DDIMScheduler
    acp = D.alphas_cumprod()
    g = torch.Generator().manual_seed(1)
    x, eps, z = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    t, n = 500, 50
    a_t, a_p = acp[t], acp[t - 1000 // n]
    det = D.ddim_step(eps, t, x, acp, n)
    assert torch.equal(D.ddim_step(eps, t, x, acp, n, eta=0.0, noise=z), det)
    for eta in (0.5, 1.0):
        sigma = eta * (((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)).sqrt()
        out = D.ddim_step(eps, t, x, acp, n, eta=eta, noise=z)
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        assert torch.allclose(out, a_p.sqrt() * x0 + (1 - a_p - sigma ** 2).sqrt() * eps + sigma * z, atol=1e-6)
        # direction^2 + sigma^2 == 1 - a_prev: the marginal q(x_{t-1} | x0) is preserved
        assert abs(float((1 - a_p - sigma ** 2) + sigma ** 2 - (1 - a_p))) < 1e-7
        sch = DDIMScheduler()
        sch.set_timesteps(n)
        c = sch.step_coefficients(t, eta)
        assert abs(float(c[4]) - float(sigma)) < 1e-6 and abs(float(c[3]) - float((1 - a_p - sigma ** 2).sqrt())) < 1e-6
    # v-prediction goes through the same variance term
    v = D.ddim_step(eps, t, x, acp, n, prediction_type='v_prediction', eta=1.0, noise=z)
    v0 = D.ddim_step(eps, t, x, acp, n, prediction_type='v_prediction', eta=1.0, noise=torch.zeros_like(z))
    sigma1 = (((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)).sqrt()
    assert torch.allclose(v - v0, sigma1 * z, atol=1e-6)
