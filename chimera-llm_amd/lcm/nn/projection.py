"""`ProjectionConfig` under the reference's module path (lcm/nn/projection.py:23-38).  The projection itself (a Linear
with an optional activation) is a product of the HIP library with the activation in its epilogue."""

from lcm.nn.transformer_config import ProjectionConfig  # noqa: F401

__all__ = ["ProjectionConfig"]
