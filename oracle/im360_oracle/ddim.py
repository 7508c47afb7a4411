"""DDIM scheduler arithmetic (diffusers/schedulers/scheduling_ddim.py:77-110, 153-209, 235-249,
251-373) for the reference's configuration: linear betas 0.00085..0.012, zero-terminal-SNR
rescale, v-prediction, eta=0, steps_offset=1, set_alpha_to_one.  TEST INFRASTRUCTURE."""
import numpy as np
import torch


def alphas_cumprod(num_train=1000, beta_start=0.00085, beta_end=0.012, zero_snr=True):
    betas = torch.linspace(beta_start, beta_end, num_train, dtype=torch.float32)
    if zero_snr:
        abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
        a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
        abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
        abar = abar_sqrt ** 2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        betas = 1 - alphas
    return torch.cumprod(1.0 - betas, dim=0)


def timesteps(n, num_train=1000, offset=1):
    ratio = num_train // n
    return torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)) + offset


def step_v(model_output, t, sample, acp, n, num_train=1000):
    """One eta=0 v-prediction update x_t -> x_{t-1}."""
    t = int(t)
    prev = t - num_train // n
    a_t = acp[t]
    a_prev = acp[prev] if prev >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
    eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
