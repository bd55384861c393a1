"""Hyper-parameter containers of the transformer stacks.

Field names and defaults follow the reference's `TransformerConfig`
(lcm/nn/transformer/factory.py:47-123), `DenoiserConfig` (lcm/nn/denoisers/factory.py:32-46),
`ProjectionConfig` (lcm/nn/projection.py:23-38) and `EncoderFrontendConfig`
(lcm/models/two_tower_diffusion_lcm/frontend.py:24-36) so that model cards with `model_config`
overrides written for the reference keep loading.
"""

from dataclasses import dataclass, field
from typing import Literal, Optional


@dataclass
class ProjectionConfig:
    dropout_p: float = 0.0
    linear_bias: bool = True
    linear_init_fn: str = "kaiming_uniform"
    weight_normalization: bool = False
    layer_normalization_style: str = "standard"
    activation_name: Optional[str] = None


@dataclass
class EncoderFrontendConfig:
    dropout_p: float = 0.0
    pre_linear_bias: bool = True
    pre_linear_init_fn: str = "kaiming_uniform"
    weight_normalization: bool = False
    embedding_std: float = 1.0


@dataclass
class TransformerConfig:
    num_layers: int = 2
    num_attn_heads: int = 8
    dropout_p: float = 0.1
    final_dropout_p: float = 0.1
    attention_dropout_p: float = 0.0
    ffn_inner_dim: int = 1024 * 4
    use_swiglu: bool = False
    ffn_inner_activation_name: str = "relu"
    pos_embedding_style: Literal["rope", "sine", "learned", "none"] = "learned"
    rope_theta: float = 10_000.0
    layer_normalization_style: Literal["standard", "fp32", "rms", "unit"] = "standard"
    norm_order_style: Literal["pre", "post", "normformer"] = "pre"
    final_norm_order_style: Optional[Literal["pre", "post", "normformer"]] = None
    enable_qk_layernorm: bool = False
    mha_qkv_weight_normalization: bool = False
    mha_output_weight_normalization: bool = False
    mha_output_proj_bias: bool = False
    scale_residual: Optional[float] = None
    attention_output_init_fn: str = "xavier"

    def allocated_ffn_dim(self) -> int:
        """Inner width of the FFN as allocated: SwiGLU scales by 2/3 and rounds up to a multiple
        of 256 (the GLUFeedForwardNetwork call at factory.py:277-285)."""
        if not self.use_swiglu:
            return self.ffn_inner_dim
        d = int(self.ffn_inner_dim * 2 / 3)
        return 256 * ((d + 255) // 256)


@dataclass
class DenoiserConfig(TransformerConfig):
    pos_embedding_style: Literal["rope", "sine", "learned", "none"] = "none"
    pre_denoiser: ProjectionConfig = field(default_factory=ProjectionConfig)
    post_denoiser: ProjectionConfig = field(default_factory=ProjectionConfig)
    timestep_embed_dim: int = 1024
