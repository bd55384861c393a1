"""Weight-normalized Linears in checkpoints.

The reference can wrap the attention projections (`mha_qkv_weight_normalization`, `mha_output_weight_normalization`:
lcm/nn/transformer/attention.py:123-127, lcm/nn/transformer/factory.py:253-254) and the frontend / postnet
projections (lcm/nn/projection.py) in `torch.nn.utils.parametrizations.weight_norm`.  Such a module has no `weight`
entry in its state dict but `parametrizations.weight.original0` (g, one magnitude per output row) and
`parametrizations.weight.original1` (v, the direction), with weight = g * v / ||v|| taken per output row (dim 0).
At inference that is a constant, so it is folded once when a state dict is loaded and the kernels see a plain
weight matrix.  The names of the deprecated `torch.nn.utils.weight_norm` (`weight_g`, `weight_v`) are accepted too.
"""

from typing import Dict

import torch
from torch import Tensor

_NEW = (".parametrizations.weight.original0", ".parametrizations.weight.original1")
_OLD = (".weight_g", ".weight_v")


def fold_weight_norm(state_dict: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Returns a state dict in which every (g, v) pair is replaced by `<module>.weight`; other entries are kept."""
    out = dict(state_dict)
    for g_suffix, v_suffix in (_NEW, _OLD):
        for key in [k for k in state_dict if k.endswith(g_suffix)]:
            prefix = key[: -len(g_suffix)]
            v_key = prefix + v_suffix
            if v_key not in state_dict:
                raise RuntimeError(f"weight-normalized module {prefix}: {key} without {v_key}")
            g, v = state_dict[key].float(), state_dict[v_key].float()
            if g.numel() != v.shape[0]:
                raise RuntimeError(f"weight-normalized module {prefix}: expected one magnitude per output row, got {tuple(g.shape)}")
            norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            out[prefix + ".weight"] = (v * (g.reshape(-1, *([1] * (v.dim() - 1))) / norm)).to(state_dict[v_key].dtype)
            del out[key], out[v_key]
    return out
