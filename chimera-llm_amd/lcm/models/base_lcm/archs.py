"""Named base-LCM architectures.

`toy_base_lcm` follows the reference's lcm/models/base_lcm/archs.py:17-21.  `base_lcm_1_6B` is NOT defined in
the reference fork (its archs file holds only the toy and an unrelated arch); it is the upstream-LCM-style
1.6B decoder used for the "base_lcm 1.6B forward" benchmark shape: d = 2048, 32 layers, 16 heads, RoPE, SwiGLU,
RMSNorm, output-projection bias.
"""

from lcm.models.base_lcm.builder import BaseLCModelConfig, lcm_arch
from lcm.nn.transformer_config import TransformerConfig


@lcm_arch("toy_base_lcm")
def toy_base_lcm() -> BaseLCModelConfig:
    return BaseLCModelConfig(lcm=TransformerConfig(num_layers=2))


@lcm_arch("base_lcm_1_6B")
def base_lcm_1_6B() -> BaseLCModelConfig:
    return BaseLCModelConfig(
        model_dim=2048, max_seq_len=4096, sonar_normalizer_name="dummy_sonar_normalizer",
        lcm=TransformerConfig(num_layers=32, num_attn_heads=16, ffn_inner_dim=4 * 2048, use_swiglu=True,
                              layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True,
                              final_dropout_p=0.0, attention_dropout_p=0.0))
