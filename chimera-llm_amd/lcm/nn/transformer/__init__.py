"""Import path of the reference (lcm/nn/transformer/__init__.py).  Only the configuration type exists here: the
reference's attention / decoder modules and its factory build torch modules, whose arithmetic lives in the HIP library
(`lcmhip_context_append`, `lcmhip_base_forward`)."""

from lcm.nn.transformer.factory import TransformerConfig  # noqa: F401

__all__ = ["TransformerConfig"]
