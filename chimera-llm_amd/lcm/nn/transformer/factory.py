"""`TransformerConfig` under the reference's module path (lcm/nn/transformer/factory.py:30-125)."""

from lcm.nn.transformer_config import TransformerConfig  # noqa: F401

__all__ = ["TransformerConfig"]
