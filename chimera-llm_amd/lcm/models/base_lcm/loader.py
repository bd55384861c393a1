"""Builds a base LCM from a card (reference: lcm/models/base_lcm/loader.py:19-55; the `module.` prefix of DDP
checkpoints is stripped as at :24-30)."""

import torch

from lcm.models.base_lcm.builder import BaseLCModel, BaseLCModelConfig, lcm_archs
from lcm.models.two_tower_diffusion_lcm.loader import _apply_overrides, convert_lcm_checkpoint  # noqa: F401


def load_base_lcm_config(name_or_card) -> BaseLCModelConfig:
    """The configuration a card describes: its `model_arch` with the card's `model_config` overrides applied
    (the reference's StandardModelConfigLoader, loader.py:33-37)."""
    from lcm.utils.card_utils import resolve_card

    card = resolve_card(name_or_card)
    config: BaseLCModelConfig = lcm_archs.get_config(card.get("model_arch", "toy_base_lcm"))
    if card.get("model_config"):
        _apply_overrides(config, card["model_config"])
    return config


load_base_lcm_config._arch_configs = lcm_archs      # where fairseq2's StandardModelConfigLoader keeps its arch registry


def load_base_lcm_model(name_or_card, device=None, dtype=None) -> BaseLCModel:
    from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model
    from lcm.utils.card_utils import resolve_card, resolve_checkpoint_path

    card = resolve_card(name_or_card)
    config = load_base_lcm_config(card)
    model = BaseLCModel(config, device=device, dtype=dtype, seed=card.get("seed"))
    ckpt = resolve_checkpoint_path(card)
    has_normalizer = False
    if ckpt is not None:
        blob = torch.load(ckpt, map_location="cpu", weights_only=True)
        sd = blob["model"] if "model" in blob else blob
        has_normalizer = any(k.endswith("sonar_normalizer.center") for k in sd)
        model.load_state_dict(sd, strict=has_normalizer or config.sonar_normalizer_name is None)
    if config.sonar_normalizer_name is not None and not has_normalizer:
        normalizer = load_sonar_normalizer_model(config.sonar_normalizer_name, device=model.device, dtype=model.dtype)
        model.sonar_normalizer.load_state_dict(normalizer.state_dict())
    return model
