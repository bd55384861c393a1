"""Import path of the reference (lcm/nn/denoisers/__init__.py).  Only the configuration type exists here: the denoiser
itself is `lcmhip_denoise` / `lcmhip_denoise_positions`."""

from lcm.nn.denoisers.factory import DenoiserConfig  # noqa: F401

__all__ = ["DenoiserConfig"]
