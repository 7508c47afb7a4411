"""DDIMScheduler with the reference's constructor keywords, attributes and ``step`` contract
(diffusers/schedulers/scheduling_ddim.py:113-373) for the configuration of configs/prompt-dual.yaml:48-56.

Coefficients are tabulated on the host in fp32 exactly like the reference (so ``alphas_cumprod`` matches
bit for bit) and combined in fp64 python floats; on the GPU the whole CFG + update chain of the
pipeline is one elementwise HIP kernel (``fused_cfg_step``).
"""
from dataclasses import dataclass

import numpy as np
import torch

from . import kernels


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class _Cfg(dict):
    __getattr__ = dict.get


def rescale_zero_terminal_snr(betas):
    """scheduling_ddim.py:77-110 (arXiv 2305.08891, algorithm 1)."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", rescale_betas_zero_snr=False, **kwargs):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, prediction_type=prediction_type,
                           rescale_betas_zero_snr=rescale_betas_zero_snr)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self._timesteps_host = [int(t) + self.config.steps_offset for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device) + self.config.steps_offset

    def _alphas(self, timestep):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def coefficients(self, timestep):
        """x_prev = cx * x_t + cv * model_output for eta = 0 (v-prediction or epsilon)."""
        a_t, a_prev = self._alphas(timestep)
        b_t = 1.0 - a_t
        if self.config.prediction_type == "v_prediction":
            cx = a_prev ** 0.5 * a_t ** 0.5 + (1 - a_prev) ** 0.5 * b_t ** 0.5
            cv = -(a_prev ** 0.5) * b_t ** 0.5 + (1 - a_prev) ** 0.5 * a_t ** 0.5
        elif self.config.prediction_type == "epsilon":
            cx = a_prev ** 0.5 / a_t ** 0.5
            cv = -(a_prev ** 0.5) * b_t ** 0.5 / a_t ** 0.5 + (1 - a_prev) ** 0.5
        else:
            raise ValueError(f"prediction_type {self.config.prediction_type} unsupported")
        return cx, cv

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if eta != 0.0 or self.config.clip_sample:
            raise NotImplementedError("the dual pipeline runs eta = 0, clip_sample = False (prompt-dual.yaml:48-56)")
        cx, cv = self.coefficients(timestep)
        prev = cx * sample + cv * model_output
        return DDIMSchedulerOutput(prev_sample=prev.to(sample.dtype)) if return_dict else (prev,)

    def fused_cfg_step(self, pred_uncond, pred_text, guidance_scale, timestep, sample, coef_dev=None):
        """CFG combine + update in one HIP kernel (pipeline_animation_inference_dual.py:791-800).  ``coef_dev``:
        device float32[3] (guidance, cx, cv) read by the kernel instead of host scalars (graph replay)."""
        cx, cv = (0.0, 0.0) if coef_dev is not None else self.coefficients(timestep)
        return kernels.cfg_ddim_update(pred_uncond.contiguous(), pred_text.contiguous(), sample.contiguous(),
                                       guidance_scale, cx, cv, coef_dev=coef_dev)
