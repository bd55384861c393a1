"""Constants of the reference's lcm/nn/initialization.py that configurations refer to (:15-25), and the activation
names its `parse_activation_fn` (:121-151) accepts, with the codes the HIP epilogues know them by."""

from typing import Literal, Optional

from lcm import _hip

SUPPORTED_INIT_TYPES = Literal["xavier", "sonar", "zero", "trunc_normal", "kaiming_uniform", "none"]

SONAR_STD = 0.006

ACTIVATION_NAMES = tuple(_hip.ACT)


def activation_code(name: Optional[str]) -> Optional[int]:
    """LCMHIP_ACT_* of an activation name (None for None); ValueError for a name the reference does not know either."""
    if name is None:
        return None
    if name not in _hip.ACT:
        raise ValueError(f"Unknown activation function {name}")
    return _hip.ACT[name]
