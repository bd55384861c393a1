"""Host half of the DDIM scheduler: noise tables and the inference timestep list.

The tensor half (get_epsilon / classifier-free guidance / step) runs in the fused HIP kernel
`lcmhip_ddim_cfg_step`; `step()` / `get_epsilon()` below call it for GPU tensors so that the
reference's per-call API (lcm/nn/schedulers/ddim.py:358-510, 561-587) keeps working.
Tables follow ddim.py:135-207 (cosine / linear betas, zero-terminal-SNR rescale, cumprod) and
the timestep list ddim.py:282-356, evaluated with the same fp32 torch expressions so that the
values are bit-identical (the truncating integer `arange` of the "trailing" spacing included).
"""

import math
from dataclasses import dataclass
from typing import Literal, NamedTuple, Optional

import torch
from torch import Tensor

from lcm import _hip


class DDIMSchedulerOutput(NamedTuple):
    prev_sample: Tensor
    pred_original_sample: Tensor


@dataclass
class DDIMSchedulerConfig:
    num_diffusion_train_steps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: Literal["linear", "scaled_linear", "squaredcos_cap_v2", "sigmoid"] = "squaredcos_cap_v2"
    scaled_linear_exponent: float = 2.0
    sigmoid_schedule_alpha: float = 1.5
    sigmoid_schedule_beta: float = 0
    clip_sample: bool = False
    clip_sample_range: float = 1.0
    set_alpha_to_one: bool = True
    prediction_type: Literal["sample", "epsilon", "v_prediction"] = "sample"
    thresholding: bool = False
    dynamic_thresholding_ratio: float = 0.995
    sample_max_value: float = 1.0
    rescale_betas_zero_snr: bool = True
    timestep_spacing: Literal["linspace", "leading", "trailing"] = "trailing"


def _cosine_betas(n: int, max_beta: float = 0.999) -> Tensor:
    bar = [math.cos((i / n + 0.008) / 1.008 * math.pi / 2) ** 2 for i in range(n + 1)]
    return torch.tensor([min(1 - bar[i + 1] / bar[i], max_beta) for i in range(n)], dtype=torch.float32)


def _sigmoid_betas(n: int, alpha: float, beta: float, max_beta: float = 0.999) -> Tensor:
    def bar(t: float) -> float:
        # the clamp and the odds run in fp32 (the reference clamps a torch scalar), the rest in Python floats;
        # t = 1 rounds to odds = inf and alpha_bar = 0
        x = torch.tensor(t).clamp(min=1e-32, max=1 - 1e-32)
        odds = float(x / (1 - x))
        z = beta - alpha * (math.inf if odds == math.inf else math.log(odds))
        return 0.0 if z == -math.inf else 1 / (1 + math.exp(-z))

    return torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


def _zero_terminal_snr(betas: Tensor) -> Tensor:
    root = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = root[0].clone(), root[-1].clone()
    root -= last
    root *= first / (first - last)
    bar = root**2
    return 1 - torch.cat([bar[0:1], bar[1:] / bar[:-1]])


def betas_for_alpha_bar(num_diffusion_timesteps: int, max_beta: float = 0.999,
                        alpha_transform_type: Literal["cosine", "exp", "sigmoid"] = "cosine",
                        sigmoid_alpha: float = 1.5, sigmoid_beta: float = 0) -> Tensor:
    """Betas that discretize an alpha-bar function of t in [0, 1] (reference ddim.py:652-706): the cosine, sigmoid
    and exponential transforms under the reference's name and arguments."""
    n = num_diffusion_timesteps
    if alpha_transform_type == "cosine":
        return _cosine_betas(n, max_beta)
    if alpha_transform_type == "sigmoid":
        return _sigmoid_betas(n, sigmoid_alpha, sigmoid_beta, max_beta)
    if alpha_transform_type == "exp":
        bar = [math.exp(i / n * -12.0) for i in range(n + 1)]
        return torch.tensor([min(1 - bar[i + 1] / bar[i], max_beta) for i in range(n)], dtype=torch.float32)
    raise ValueError(f"Unsupported alpha_transform_type: {alpha_transform_type}")


def rescale_zero_terminal_snr(betas: Tensor) -> Tensor:
    """Betas rescaled to zero terminal SNR (reference ddim.py:709-741; arXiv 2305.08891, algorithm 1)."""
    return _zero_terminal_snr(betas)


class DDIMScheduler:
    def __init__(self, config: DDIMSchedulerConfig):
        self.config = config
        self.num_diffusion_train_steps = n = config.num_diffusion_train_steps
        self.prediction_type = config.prediction_type
        kind = config.beta_schedule
        if kind == "linear":
            betas = torch.linspace(config.beta_start, config.beta_end, n, dtype=torch.float32)
        elif kind == "scaled_linear":
            e = config.scaled_linear_exponent
            betas = torch.linspace(config.beta_start ** (1 / e), config.beta_end ** (1 / e), n, dtype=torch.float32) ** e
        elif kind == "squaredcos_cap_v2":
            betas = _cosine_betas(n)
        elif kind == "sigmoid":
            betas = _sigmoid_betas(n, config.sigmoid_schedule_alpha, config.sigmoid_schedule_beta)
        else:
            raise NotImplementedError(f"We do not recognize beta_schedule={kind}")
        if config.rescale_betas_zero_snr:
            betas = _zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if config.set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps: Optional[Tensor] = None

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n = self.num_diffusion_train_steps
        if num_inference_steps > n:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                f"`self.config.num_diffusion_train_steps`: {n}")
        self.num_inference_steps = num_inference_steps
        spacing = self.config.timestep_spacing
        if spacing == "trailing":
            ts = torch.arange(n, 0, -(n / num_inference_steps), dtype=torch.long).round() - 1
        elif spacing == "leading":
            ts = torch.arange(0, n, n // num_inference_steps, dtype=torch.long).flip(0).round()
        elif spacing == "linspace":
            ts = torch.linspace(0, n - 1, num_inference_steps, dtype=torch.long).flip(0).round()
        else:
            raise ValueError(f"{spacing} is not supported. Please make sure to choose one of 'leading' or 'trailing'.")
        self.timesteps = ts  # host tensor: the schedule is compiled into the session, never read on device

    # ---- scalar helpers -------------------------------------------------------------------------
    @property
    def step_ratio(self) -> int:
        return self.num_diffusion_train_steps // self.num_inference_steps

    def alpha_pair(self, timestep: int):
        prev = int(timestep) - self.step_ratio
        a_t = float(self.alphas_cumprod[int(timestep)])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    # ---- host tables of the reference's analysis helpers (ddim.py:209-240) ---------------------------------------
    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def get_variances(self) -> Tensor:
        """The DDIM variance of every training step against its predecessor (the first against `final_alpha_cumprod`)."""
        a_t = self.alphas_cumprod
        a_prev = torch.cat((torch.tensor([self.final_alpha_cumprod]), a_t[:-1]))
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def get_snrs(self) -> Tensor:
        return self.alphas_cumprod / (1 - self.alphas_cumprod)

    def _params(self, eta, epsilon_scaling, prediction_type, clip_noise=float("inf")) -> "_hip.DenoiseParams":
        return _hip.DenoiseParams(
            guidance_scale=1.0, guidance_rescale=0.0, ddim_eta=float(eta),
            epsilon_scaling=float(epsilon_scaling) if epsilon_scaling else 0.0,
            clip_noise=float(min(clip_noise, 3.0e38)), init_noise_sigma=float(self.init_noise_sigma),
            prediction_type=_hip.PRED[prediction_type or self.prediction_type],
            clip_sample=int(self.config.clip_sample), clip_sample_range=float(self.config.clip_sample_range),
            thresholding=int(self.config.thresholding),
            dynamic_thresholding_ratio=float(self.config.dynamic_thresholding_ratio),
            sample_max_value=float(self.config.sample_max_value))

    # ---- per-call tensor API (GPU tensors, fp32) ---------------------------------------------------
    def step(self, model_output: Tensor, timestep, sample: Tensor, eta: float = 0.0,
             prediction_type: Optional[str] = None, epsilon_scaling: Optional[float] = None,
             variance_noise: Optional[Tensor] = None, generator=None) -> DDIMSchedulerOutput:
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        _hip.require_gpu(sample, "sample")
        x = sample.float().contiguous()
        out = model_output.float().contiguous()
        dim = x.shape[-1]
        rows = x.numel() // dim
        prev, x0 = torch.empty_like(x), torch.empty_like(x)
        a_t, a_prev = self.alpha_pair(int(timestep))
        p = self._params(eta, epsilon_scaling, prediction_type)
        vn = None
        if eta > 0:                                   # reference ddim.py:487-506
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = torch.randn(x.shape, generator=generator, device=x.device, dtype=torch.float32)
            vn = variance_noise.float().contiguous()
            assert vn.numel() == x.numel(), "variance_noise must have the shape of the sample"
        _hip.check(_hip.lib.lcmhip_ddim_cfg_step(_hip.ptr(x), _hip.ptr(out), None, _hip.ptr(prev), _hip.ptr(x0),
                                                 rows, dim, a_t, a_prev, p, _hip.ptr(vn), _hip.stream_ptr(x.device)))
        return DDIMSchedulerOutput(prev, x0)

    def _rows_by_timestep(self, first: Tensor, second: Tensor, timesteps: Tensor, alpha_t_of, alpha_prev_of,
                          prediction_type: str, want_x0: bool) -> Tensor:
        """One deterministic step-kernel call per distinct timestep over the rows that carry it (rows are sorted by
        timestep first, so every call sees a contiguous block): `first` is the kernel's sample, `second` its model
        output; returns prev_sample or the predicted x0, fp32, in the order of the input rows."""
        _hip.require_gpu(first, "sample")
        x, y = first.float().contiguous(), second.float().contiguous()
        assert x.shape == y.shape, (tuple(x.shape), tuple(y.shape))
        dim = x.shape[-1]
        x2, y2 = x.reshape(-1, dim), y.reshape(-1, dim)
        t = timesteps.to(x.device).long().flatten()
        assert t.numel() >= 1 and x2.shape[0] % t.numel() == 0, (tuple(x.shape), tuple(timesteps.shape))
        t = t.repeat_interleave(x2.shape[0] // t.numel())       # one timestep per leading entry of the samples
        order = torch.argsort(t, stable=True)
        xs, ys = x2[order].contiguous(), y2[order].contiguous()
        steps, counts = torch.unique_consecutive(t[order], return_counts=True)
        res = torch.empty_like(xs)
        other = torch.empty_like(xs)
        p = _hip.DenoiseParams(
            guidance_scale=1.0, guidance_rescale=0.0, ddim_eta=0.0, epsilon_scaling=0.0, clip_noise=3.0e38,
            init_noise_sigma=1.0, prediction_type=_hip.PRED[prediction_type], clip_sample=0, clip_sample_range=1.0,
            thresholding=0, dynamic_thresholding_ratio=0.995, sample_max_value=1.0)
        st, es, row0 = _hip.stream_ptr(x.device), xs.element_size() * dim, 0
        for step, n in zip(steps.tolist(), counts.tolist()):
            a = float(self.alphas_cumprod[int(step)])
            off = row0 * es
            prev, x0 = (other, res) if want_x0 else (res, other)
            _hip.check(_hip.lib.lcmhip_ddim_cfg_step(xs.data_ptr() + off, ys.data_ptr() + off, None, prev.data_ptr() + off,
                                                     x0.data_ptr() + off, n, dim, alpha_t_of(a), alpha_prev_of(a), p, None, st))
            row0 += n
        out = torch.empty_like(res)
        out[order] = res
        return out.reshape(x.shape)

    def add_noise(self, original_samples: Tensor, noise: Tensor, timesteps: Tensor) -> Tensor:
        """The forward noising process sqrt(a_t) x + sqrt(1 - a_t) noise with one timestep per leading entry of the
        samples (reference ddim.py:512-540; a training-side member of the class), fp32.  Computed by the step kernel:
        an epsilon-type deterministic step from alpha = 1 (where x0 is the sample itself) to alpha_prev = a_t with the
        noise as the model output is sqrt(a_t) x + sqrt(1 - a_t) noise."""
        return self._rows_by_timestep(original_samples, noise, timesteps, lambda a: 1.0, lambda a: a, "epsilon", False)

    def get_velocity(self, sample: Tensor, noise: Tensor, timesteps: Tensor) -> Tensor:
        """sqrt(a_t) noise - sqrt(1 - a_t) sample (reference ddim.py:542-559), fp32: the x0 that the step kernel
        predicts for a v-type model output `sample` at the latent `noise`."""
        return self._rows_by_timestep(noise, sample, timesteps, lambda a: a, lambda a: 0.0, "v_prediction", True)

    def get_epsilon(self, model_output: Tensor, sample: Tensor, timestep) -> Tensor:
        """The noise the model output implies for `sample` at `timestep` under the scheduler's prediction type
        (reference ddim.py:561-587), fp32.  Computed by the step kernel: with alpha_prev = 0 a deterministic DDIM step
        returns sqrt(0) x0 + sqrt(1) epsilon, i.e. the predicted noise itself."""
        if self.prediction_type == "epsilon":
            return model_output
        if self.prediction_type not in ("sample", "v_prediction"):
            raise ValueError(
                f"prediction_type given as {self.prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        _hip.require_gpu(sample, "sample")
        x = sample.float().contiguous()
        out = model_output.float().contiguous()
        assert out.shape == x.shape, (tuple(out.shape), tuple(x.shape))
        dim = x.shape[-1]
        eps = torch.empty_like(x)
        p = _hip.DenoiseParams(
            guidance_scale=1.0, guidance_rescale=0.0, ddim_eta=0.0, epsilon_scaling=0.0, clip_noise=3.0e38,
            init_noise_sigma=1.0, prediction_type=_hip.PRED[self.prediction_type], clip_sample=0, clip_sample_range=1.0,
            thresholding=0, dynamic_thresholding_ratio=0.995, sample_max_value=1.0)
        _hip.check(_hip.lib.lcmhip_ddim_cfg_step(_hip.ptr(x), _hip.ptr(out), None, _hip.ptr(eps), None, x.numel() // dim, dim,
                                                 float(self.alphas_cumprod[int(timestep)]), 0.0, p, None,
                                                 _hip.stream_ptr(x.device)))
        return eps
