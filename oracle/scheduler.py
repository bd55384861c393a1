"""DDIM scheduler restatement (test infrastructure).

Follows lcm/nn/schedulers/ddim.py: tables :135-207, `set_timesteps` :282-356, `step` :358-510,
`get_epsilon` :561-587, `betas_for_alpha_bar` :652-706, `rescale_zero_terminal_snr` :709-741.
All table arithmetic is fp32 torch on the CPU, like the reference's.
"""

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from .archs import SchedulerConfig


def betas_for_alpha_bar(n: int, max_beta: float = 0.999, transform: str = "cosine", sigmoid_alpha: float = 1.5,
                        sigmoid_beta: float = 0.0) -> Tensor:
    # ddim.py:652-706: "cosine" (the archs on the path use squaredcos_cap_v2) and "sigmoid"
    if transform == "cosine":
        def alpha_bar(t: float) -> float:
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    elif transform == "sigmoid":
        # ddim.py:20-27, 683-691: t is clamped as an fp32 torch scalar and x / (1 - x) is an fp32 tensor; math.log and
        # math.exp then run on its Python float.  t = 1 clamps to 1.0 in fp32: odds inf, alpha_bar 0.
        def alpha_bar(t: float) -> float:
            x = torch.clamp(torch.tensor(t), min=1e-32, max=1 - 1e-32)
            return 1 / (1 + math.exp(-(sigmoid_beta - sigmoid_alpha * math.log(x / (1 - x)))))
    else:
        raise ValueError(f"Unsupported alpha_transform_type: {transform}")

    betas = [min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)]
    return torch.tensor(betas, dtype=torch.float32)


def rescale_zero_terminal_snr(betas: Tensor) -> Tensor:
    # ddim.py:709-741
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= aT
    alphas_bar_sqrt *= a0 / (a0 - aT)
    alphas_bar = alphas_bar_sqrt**2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1 - alphas


class DDIMSchedule:
    def __init__(self, cfg: SchedulerConfig):
        self.cfg = cfg
        T = cfg.num_train_steps
        self.num_train_steps = T
        self.prediction_type = cfg.prediction_type
        if cfg.beta_schedule == "linear":
            betas = torch.linspace(cfg.beta_start, cfg.beta_end, T, dtype=torch.float32)
        elif cfg.beta_schedule == "scaled_linear":       # ddim.py:153-164
            e = cfg.scaled_linear_exponent
            betas = torch.linspace(cfg.beta_start ** (1 / e), cfg.beta_end ** (1 / e), T, dtype=torch.float32) ** e
        elif cfg.beta_schedule == "squaredcos_cap_v2":
            betas = betas_for_alpha_bar(T)
        elif cfg.beta_schedule == "sigmoid":             # ddim.py:173-179
            betas = betas_for_alpha_bar(T, transform="sigmoid", sigmoid_alpha=cfg.sigmoid_schedule_alpha,
                                        sigmoid_beta=cfg.sigmoid_schedule_beta)
        else:
            raise NotImplementedError(cfg.beta_schedule)
        if cfg.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps: Optional[Tensor] = None

    def set_timesteps(self, num_inference_steps: int) -> None:
        # ddim.py:282-356.  The integer dtype truncates the fractional step of the "trailing"
        # arange (T=100, S=40 -> step -2 -> 50 timesteps), so the identical torch expression is
        # evaluated instead of re-deriving the list.
        T = self.num_train_steps
        if num_inference_steps > T:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {T}")
        self.num_inference_steps = num_inference_steps
        sp = self.cfg.timestep_spacing
        if sp == "linspace":
            ts = torch.flip(torch.linspace(0, T - 1, num_inference_steps, dtype=torch.long), dims=(0,)).round()
        elif sp == "leading":
            ts = torch.flip(torch.arange(0, T, T // num_inference_steps, dtype=torch.long), dims=(0,)).round()
        elif sp == "trailing":
            ts = torch.arange(T, 0, -(T / num_inference_steps), dtype=torch.long).round()
            ts -= 1
        else:
            raise ValueError(sp)
        self.timesteps = ts

    def _alphas(self, timestep: int) -> Tuple[Tensor, Tensor]:
        prev = timestep - self.num_train_steps // self.num_inference_steps          # ddim.py:421-423
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod   # :427-431
        return a_t, a_prev

    def get_epsilon(self, model_output: Tensor, sample: Tensor, timestep: int) -> Tensor:
        # ddim.py:561-587
        a_t = self.alphas_cumprod[timestep]
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            return model_output
        if self.prediction_type == "sample":
            return (sample - a_t**0.5 * model_output) / b_t**0.5
        if self.prediction_type == "v_prediction":
            return a_t**0.5 * model_output + b_t**0.5 * sample
        raise ValueError(self.prediction_type)

    def step(self, model_output: Tensor, timestep: int, sample: Tensor, eta: float = 0.0,
             prediction_type: Optional[str] = None, epsilon_scaling: Optional[float] = None,
             variance_noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """Returns (prev_sample, pred_original_sample); ddim.py:358-510."""
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        a_t, a_prev = self._alphas(timestep)
        b_t = 1 - a_t
        ptype = prediction_type or self.prediction_type
        if ptype == "epsilon":
            x0 = (sample - b_t**0.5 * model_output) / a_t**0.5
            eps = model_output
        elif ptype == "sample":
            x0 = model_output
            eps = (sample - a_t**0.5 * x0) / b_t**0.5
        elif ptype == "v_prediction":
            x0 = a_t**0.5 * sample - b_t**0.5 * model_output
            eps = a_t**0.5 * model_output + b_t**0.5 * sample
        else:
            raise ValueError(ptype)
        if epsilon_scaling is not None:
            eps = eps / epsilon_scaling
        if self.cfg.thresholding:
            x0 = self._threshold(x0)
        elif self.cfg.clip_sample:
            x0 = x0.clamp(-self.cfg.clip_sample_range, self.cfg.clip_sample_range)
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)                  # ddim.py:209-222
        std = eta * variance**0.5
        prev = a_prev**0.5 * x0 + (1 - a_prev - std**2) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return prev, x0

    def _threshold(self, sample: Tensor) -> Tensor:
        # ddim.py:242-280
        shape = sample.shape
        flat = sample.float().reshape(shape[0], -1)
        s = torch.quantile(flat.abs(), self.cfg.dynamic_thresholding_ratio, dim=1)
        s = torch.clamp(s, min=1, max=self.cfg.sample_max_value).unsqueeze(1)
        return (torch.clamp(flat, -s, s) / s).reshape(shape).to(sample.dtype)
