"""`EncoderFrontendConfig` under the reference's module path (lcm/models/two_tower_diffusion_lcm/frontend.py:22-52).
The frontend itself (Linear + optional positions) is the first stage of `lcmhip_context_append`."""

from lcm.nn.transformer_config import EncoderFrontendConfig  # noqa: F401

__all__ = ["EncoderFrontendConfig"]
