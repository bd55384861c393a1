"""Architecture tables for the oracle (test infrastructure).

Values follow lcm/models/two_tower_diffusion_lcm/archs.py:17-112 and the defaults in
lcm/nn/transformer/factory.py:47-123, lcm/nn/denoisers/factory.py:32-46,
lcm/nn/schedulers/ddim.py:48-132 and lcm/models/two_tower_diffusion_lcm/builder.py:49-81.
"""

from dataclasses import dataclass, field


def glu_inner_dim(inner_dim: int, scale: float = 2 / 3, multiple: int = 256) -> int:
    """fairseq2 `GLUFeedForwardNetwork(inner_dim_scale=2/3, inner_dim_to_multiple=256)` as
    called at lcm/nn/transformer/factory.py:277-285: scale, truncate, round up."""
    d = int(inner_dim * scale)
    return multiple * ((d + multiple - 1) // multiple)


@dataclass
class TowerConfig:
    num_layers: int = 2
    num_heads: int = 8
    ffn_inner_dim: int = 4096          # as configured (before GLU rounding)
    use_swiglu: bool = False
    activation: str = "relu"
    norm: str = "standard"             # "standard" (LayerNorm w+b) | "rms"
    pos: str = "learned"               # "learned" | "rope" | "none"
    rope_theta: float = 10000.0
    out_proj_bias: bool = False

    def ffn_dim(self) -> int:
        return glu_inner_dim(self.ffn_inner_dim) if self.use_swiglu else self.ffn_inner_dim


@dataclass
class SchedulerConfig:
    num_train_steps: int = 1000
    beta_schedule: str = "squaredcos_cap_v2"
    beta_start: float = 1e-4
    beta_end: float = 0.02
    prediction_type: str = "sample"
    rescale_betas_zero_snr: bool = True
    set_alpha_to_one: bool = True
    timestep_spacing: str = "trailing"
    clip_sample: bool = False
    clip_sample_range: float = 1.0
    thresholding: bool = False
    dynamic_thresholding_ratio: float = 0.995
    sample_max_value: float = 1.0


@dataclass
class ArchConfig:
    name: str = "toy_two_tower_diffusion_lcm"
    sonar_dim: int = 1024
    model_dim: int = 1024
    max_seq_len: int = 2048
    norm_eps: float = 1e-5
    context: TowerConfig = field(default_factory=TowerConfig)
    denoiser: TowerConfig = field(default_factory=lambda: TowerConfig(pos="none"))
    timestep_freq_dim: int = 256
    scheduler: SchedulerConfig = field(default_factory=SchedulerConfig)
    trained_with_cf_guidance: bool = False
    sonar_normalizer_name: str = "dummy_sonar_normalizer"


def _toy() -> ArchConfig:
    # archs.py:17-24: everything default except the layer counts (already 2).
    return ArchConfig()


def _big(name: str, d: int, heads: int, den_layers: int) -> ArchConfig:
    ctx = TowerConfig(num_layers=5, num_heads=heads, ffn_inner_dim=4 * d, use_swiglu=True,
                      norm="rms", pos="rope", out_proj_bias=True)
    den = TowerConfig(num_layers=den_layers, num_heads=heads, ffn_inner_dim=4 * d,
                      use_swiglu=True, norm="rms", pos="none", out_proj_bias=True)
    return ArchConfig(name=name, model_dim=d, max_seq_len=4096, context=ctx, denoiser=den,
                      scheduler=SchedulerConfig(num_train_steps=100),
                      trained_with_cf_guidance=True)


def _mini() -> ArchConfig:
    """Not a reference arch: a 1.6B-*style* (rms / rope / swiglu / out-proj bias, T=100) model
    small enough for CPU unit tests.  head_dim stays 128 like every reference arch."""
    ctx = TowerConfig(num_layers=2, num_heads=2, ffn_inner_dim=1024, use_swiglu=True,
                      norm="rms", pos="rope", out_proj_bias=True)
    den = TowerConfig(num_layers=3, num_heads=2, ffn_inner_dim=1024, use_swiglu=True,
                      norm="rms", pos="none", out_proj_bias=True)
    return ArchConfig(name="mini_two_tower_diffusion_lcm", model_dim=256, max_seq_len=512,
                      context=ctx, denoiser=den,
                      scheduler=SchedulerConfig(num_train_steps=100),
                      trained_with_cf_guidance=True)


ARCHS = {
    "toy_two_tower_diffusion_lcm": _toy,
    "two_tower_diffusion_lcm_1_6B": lambda: _big("two_tower_diffusion_lcm_1_6B", 2048, 16, 13),
    "two_tower_diffusion_lcm_7B": lambda: _big("two_tower_diffusion_lcm_7B", 4096, 32, 14),
    "mini_two_tower_diffusion_lcm": _mini,
}


def get_arch(name: str) -> ArchConfig:
    cfg = ARCHS[name]()
    cfg.name = name
    return cfg
