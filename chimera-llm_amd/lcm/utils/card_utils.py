"""YAML model cards -> models (reference: lcm/utils/card_utils.py:178-223 `load_model_from_card`).

A card is a mapping with `name`, `model_family`, `model_arch`, `checkpoint` and optionally
`model_config` (overrides) and `seed`.  `checkpoint` may be `file://...`, a path relative to
the card, or null / `/dev/null` for a seeded random initialisation.
"""

import os
from pathlib import Path
from typing import Any, Dict, Optional, Union

import yaml

CARDS_DIR = Path(__file__).resolve().parent.parent / "cards"
_in_proc_cards: Dict[str, Dict[str, Any]] = {}


def _read_yaml(path: Path) -> Dict[str, Any]:
    with open(path, "r") as f:
        card = yaml.safe_load(f)
    card["__dir__"] = str(path.parent)
    return card


def register_card(card: Dict[str, Any]) -> None:
    _in_proc_cards[card["name"]] = dict(card)


def resolve_card(name_or_card: Union[str, Path, Dict[str, Any]]) -> Dict[str, Any]:
    if isinstance(name_or_card, dict):
        return name_or_card
    s = str(name_or_card)
    if s in _in_proc_cards:
        return _in_proc_cards[s]
    p = Path(s)
    if p.suffix in (".yaml", ".yml") and p.exists():
        return _read_yaml(p)
    if p.is_dir() and (p / "model_card.yaml").exists():
        return _read_yaml(p / "model_card.yaml")
    for f in sorted(CARDS_DIR.glob("*.yaml")):
        card = _read_yaml(f)
        if card.get("name") == s:
            return card
    raise ValueError(f"no model card named '{s}' (searched {CARDS_DIR} and in-process cards)")


def resolve_checkpoint_path(card: Dict[str, Any]) -> Optional[str]:
    ckpt = card.get("checkpoint")
    if ckpt is None:
        return None
    ckpt = str(ckpt)
    if ckpt.startswith("file://"):
        ckpt = ckpt[len("file://"):]
    if ckpt in ("", "/dev/null"):
        return None
    if not os.path.isabs(ckpt):
        ckpt = os.path.join(card.get("__dir__", str(CARDS_DIR)), ckpt)
    return ckpt


def create_model_card(checkpoint_path: Optional[Path], model_config: Optional[Dict[str, Any]], model_type: str,
                      model_name: str = "on_the_fly_lcm", model_arch: Optional[str] = None, **extra) -> Dict[str, Any]:
    card: Dict[str, Any] = {"name": model_name, "model_family": model_type,
                            "checkpoint": None if checkpoint_path is None else "file://" + Path(checkpoint_path).as_posix(),
                            **extra}
    if model_config is not None:
        card["model_config"] = model_config
    if model_arch is not None:
        card["model_arch"] = model_arch
    register_card(card)
    return card


def load_model_from_card(model_name_or_path: Union[str, Path, Dict[str, Any]], device=None, dtype=None):
    from lcm.utils.model_type_registry import lcm_model_type_registry

    card = resolve_card(model_name_or_path)
    loader = lcm_model_type_registry.get_model_loader(card["model_family"])
    return loader(card, device=device, dtype=dtype)
