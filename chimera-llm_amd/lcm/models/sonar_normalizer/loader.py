"""Loads a normalizer from a card (reference: lcm/models/sonar_normalizer/loader.py:14-28)."""

import torch

from lcm.models.sonar_normalizer.builder import SonarNormalizer, SonarNormalizerConfig, sonar_normalizer_archs


def load_sonar_normalizer_config(name_or_card) -> SonarNormalizerConfig:
    """The configuration a normalizer card describes (the reference's StandardModelConfigLoader, loader.py:14-18)."""
    import dataclasses

    from lcm.utils.card_utils import resolve_card

    card = resolve_card(name_or_card)
    config = dataclasses.replace(sonar_normalizer_archs[card.get("model_arch", "base")])
    for k, v in (card.get("model_config") or {}).items():
        if not hasattr(config, k):
            raise ValueError(f"model_config: unknown field '{k}' for SonarNormalizerConfig")
        setattr(config, k, v)
    return config


def load_sonar_normalizer_model(name_or_card, device=None, dtype=None) -> SonarNormalizer:
    from lcm.utils.card_utils import resolve_card, resolve_checkpoint_path

    card = resolve_card(name_or_card)
    assert card.get("model_family", "sonar_normalizer") == "sonar_normalizer"
    config = load_sonar_normalizer_config(card)
    model = SonarNormalizer(config, device=device, dtype=dtype)
    blob = torch.load(resolve_checkpoint_path(card), map_location="cpu", weights_only=True)
    model.load_state_dict(blob["model"] if "model" in blob else blob)
    return model
