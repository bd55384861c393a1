"""`LCMFrontendConfig` under the reference's module path (lcm/models/base_lcm/frontend.py:24-60).  The frontend itself
(normalizer, Linear, optional positions and embedding scale) is the first stage of `lcmhip_base_forward`."""

from lcm.models.base_lcm.builder import LCMFrontendConfig  # noqa: F401

__all__ = ["LCMFrontendConfig"]
