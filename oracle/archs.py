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
    pos: str = "learned"               # "learned" | "rope" | "sine" | "none"
    rope_theta: float = 10000.0
    out_proj_bias: bool = False
    qk_layernorm: bool = False         # LayerNorm on q and k after their projections (attention.py:143-199)
    # contextualizer / base-LCM stacks only (lcm/nn/transformer/factory.py:95-102, 120-121)
    norm_order: str = "pre"            # "pre" | "post" | "normformer"
    final_norm_order: str = None       # None: the layers' order; "post": the stack ends without its LayerNorm
    scale_residual: bool = False       # the FFN block's residual is multiplied by <layer>.residual_scale

    def ffn_dim(self) -> int:
        return glu_inner_dim(self.ffn_inner_dim) if self.use_swiglu else self.ffn_inner_dim


@dataclass
class SchedulerConfig:
    num_train_steps: int = 1000
    beta_schedule: str = "squaredcos_cap_v2"
    beta_start: float = 1e-4
    beta_end: float = 0.02
    scaled_linear_exponent: float = 2.0    # ddim.py:87
    sigmoid_schedule_alpha: float = 1.5    # ddim.py:90-91
    sigmoid_schedule_beta: float = 0.0
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
    normalizer_fft: bool = False           # the "base_fft" / "clipping_fft" normalizers (sonar_normalizer/archs.py:28-40)
    frontend_bias: bool = True             # two_tower_diffusion_lcm/frontend.py:24-36 pre_linear_bias
    initial_proj_bias: bool = True         # lcm/nn/projection.py:28 on denoiser.pre_denoiser / post_denoiser
    final_proj_bias: bool = True


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


@dataclass
class BaseArchConfig:
    """lcm/models/base_lcm/builder.py:37-57 (BaseLCModelConfig): frontend -> causal decoder -> postnet."""
    name: str = "toy_base_lcm"
    sonar_dim: int = 1024
    model_dim: int = 1024
    max_seq_len: int = 2048
    norm_eps: float = 1e-5
    lcm: TowerConfig = field(default_factory=TowerConfig)
    pre_linear_bias: bool = True           # base_lcm/frontend.py:30
    postnet_bias: bool = True              # lcm/nn/projection.py:28
    postnet_activation: str = None         # lcm/nn/projection.py:37, 73-86
    normalizer_fft: bool = False
    scale_embeddings: bool = False         # base_lcm/frontend.py:35, 91
    sonar_normalizer_name: str = None      # abstract_lcm/builder.py:33: no normalizer unless named


def _toy_base() -> BaseArchConfig:
    # lcm/models/base_lcm/archs.py:17-21: defaults with a 2-layer decoder.
    return BaseArchConfig()


def _base_1_6b() -> BaseArchConfig:
    """`base_lcm_1_6B` is NOT defined in the reference fork (its base_lcm/archs.py holds only the toy and an
    unrelated arch); this is the upstream-LCM-style reconstruction used for the "base_lcm 1.6B forward" benchmark
    shape: d = 2048, 32 layers, 16 heads, RoPE, SwiGLU (8192 -> 5632), RMSNorm, output-projection bias."""
    t = TowerConfig(num_layers=32, num_heads=16, ffn_inner_dim=8192, use_swiglu=True, norm="rms", pos="rope",
                    out_proj_bias=True)
    return BaseArchConfig(name="base_lcm_1_6B", model_dim=2048, max_seq_len=4096, lcm=t,
                          sonar_normalizer_name="dummy_sonar_normalizer")


def _mini_base() -> BaseArchConfig:
    """Not a reference arch: 1.6B-style blocks small enough for unit tests."""
    t = TowerConfig(num_layers=3, num_heads=2, ffn_inner_dim=1024, use_swiglu=True, norm="rms", pos="rope",
                    out_proj_bias=True)
    return BaseArchConfig(name="mini_base_lcm", model_dim=256, max_seq_len=512, lcm=t,
                          sonar_normalizer_name="dummy_sonar_normalizer")


BASE_ARCHS = {"toy_base_lcm": _toy_base, "base_lcm_1_6B": _base_1_6b, "mini_base_lcm": _mini_base}


def get_base_arch(name: str) -> BaseArchConfig:
    cfg = BASE_ARCHS[name]()
    cfg.name = name
    return cfg


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
