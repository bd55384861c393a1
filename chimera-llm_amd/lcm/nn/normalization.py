"""Layer-norm styles of the reference's lcm/nn/normalization.py:13-88 and the codes the HIP kernels know them by
(`lcmhip_tower_desc.norm_kind`)."""

from typing import Literal

from lcm import _hip

SUPPORTED_LN_TYPES = Literal["standard", "fp32", "rms", "unit"]


NORM_KIND = {"standard": _hip.NORM_LAYERNORM, "rms": _hip.NORM_RMS, "fp32": _hip.NORM_FP32, "unit": _hip.NORM_UNIT}


def norm_kind(layer_normalization_style: str) -> int:
    """LCMHIP_NORM_* of a style name; ValueError for an unknown one (the reference's parse_layer_norm_factory raises too)."""
    if layer_normalization_style not in NORM_KIND:
        raise ValueError(f"Unsupported layer normalization style: {layer_normalization_style}")
    return NORM_KIND[layer_normalization_style]
