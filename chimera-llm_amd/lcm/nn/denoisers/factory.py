"""`DenoiserConfig` under the reference's module path (lcm/nn/denoisers/factory.py:34-60)."""

from lcm.nn.transformer_config import DenoiserConfig  # noqa: F401

__all__ = ["DenoiserConfig"]
