"""Builds a model from a card (reference: lcm/models/two_tower_diffusion_lcm/loader.py:19-44 and
lcm/models/base_lcm/loader.py:24-30 for the `module.` prefix of DDP checkpoints)."""

import dataclasses
from typing import Any, Dict

import torch

from lcm.models.two_tower_diffusion_lcm.builder import (
    TwoTowerDiffusionLCModel,
    TwoTowerDiffusionLCModelConfig,
    lcm_archs,
)


def _apply_overrides(obj: Any, overrides: Dict[str, Any]) -> Any:
    """`model_config` of a card: nested dicts override dataclass fields."""
    for key, value in overrides.items():
        if not hasattr(obj, key):
            raise ValueError(f"model_config: unknown field '{key}' for {type(obj).__name__}")
        cur = getattr(obj, key)
        if dataclasses.is_dataclass(cur) and isinstance(value, dict):
            _apply_overrides(cur, value)
        else:
            setattr(obj, key, value)
    return obj


def load_two_tower_diffusion_lcm_config(name_or_card) -> TwoTowerDiffusionLCModelConfig:
    """The configuration a card describes: its `model_arch` with the card's `model_config` overrides applied
    (the reference's StandardModelConfigLoader, loader.py:19-23)."""
    from lcm.utils.card_utils import resolve_card

    card = resolve_card(name_or_card)
    config: TwoTowerDiffusionLCModelConfig = lcm_archs.get_config(card.get("model_arch", "toy_two_tower_diffusion_lcm"))
    if card.get("model_config"):
        _apply_overrides(config, card["model_config"])
    return config


load_two_tower_diffusion_lcm_config._arch_configs = lcm_archs      # where fairseq2's StandardModelConfigLoader keeps its arch registry


def convert_lcm_checkpoint(checkpoint: Dict[str, Any], config: Any = None) -> Dict[str, Any]:
    """DDP checkpoints: the `module.` prefix of the state-dict keys is removed in place (base_lcm/loader.py:24-30)."""
    sd = checkpoint["model"]
    for k in [k for k in sd if k.startswith("module.")]:
        sd[k[len("module."):]] = sd.pop(k)
    return checkpoint


def load_two_tower_diffusion_lcm_model(name_or_card, device=None, dtype=None) -> TwoTowerDiffusionLCModel:
    from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model
    from lcm.utils.card_utils import resolve_card, resolve_checkpoint_path

    card = resolve_card(name_or_card)
    config = load_two_tower_diffusion_lcm_config(card)
    model = TwoTowerDiffusionLCModel(config, device=device, dtype=dtype, seed=card.get("seed"))
    ckpt = resolve_checkpoint_path(card)
    if ckpt is not None:
        blob = torch.load(ckpt, map_location="cpu", weights_only=True)
        sd = blob["model"] if "model" in blob else blob
        # checkpoints written before the normalizer was part of the model do not carry its buffers
        strict = any(k.endswith("sonar_normalizer.center") for k in sd)
        model.load_state_dict(sd, strict=strict)
        if strict:
            return model
    if config.sonar_normalizer_name is not None:
        normalizer = load_sonar_normalizer_model(config.sonar_normalizer_name, device=model.device, dtype=model.dtype)
        model.sonar_normalizer.load_state_dict(normalizer.state_dict())
    return model
