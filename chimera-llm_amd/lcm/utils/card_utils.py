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


def create_model_card(checkpoint_path: Optional[Path], model_config: Any, model_type: str,
                      model_name: str = "on_the_fly_lcm", model_arch: Optional[str] = None, **extra) -> Dict[str, Any]:
    """A card made on the fly and registered in process (reference :28-69).  `model_config` may be a dict of overrides
    or a configuration dataclass."""
    import dataclasses
    if dataclasses.is_dataclass(model_config) and not isinstance(model_config, type):
        model_config = dataclasses.asdict(model_config)
    card: Dict[str, Any] = {"name": model_name, "model_family": model_type,
                            "checkpoint": None if checkpoint_path is None else "file://" + Path(checkpoint_path).as_posix(),
                            **extra}
    if model_config is not None:
        card["model_config"] = model_config
    if model_arch is not None:
        card["model_arch"] = model_arch
    register_card(card)
    return card


def save_model_card(card: Dict[str, Any], path: Union[str, Path]) -> None:
    """Writes a card as YAML (reference :171-175)."""
    data = {k: v for k, v in card.items() if not k.startswith("__")}
    with open(path, "w", encoding="utf-8") as f:
        yaml.safe_dump(data, f, default_flow_style=False)


def _step_dirs(folder: Path):
    """(step number, directory) of every `checkpoints/step_<n>` of a training folder, ascending."""
    out = []
    for d in (folder / "checkpoints").glob("step_*"):
        if d.is_dir() and d.name[5:].isdigit():
            out.append((int(d.name[5:]), d))
    return sorted(out)


def create_model_card_from_training_folder(folder: Union[str, Path], card_name: str,
                                           step_nr: Optional[int] = None) -> Dict[str, Any]:
    """A card for a checkpoint of a training folder (reference :128-168), the last step unless `step_nr` is given.
    The reference reads the model configuration from the checkpoint manager's pickled metadata; here it is read from
    the `model_card.yaml` its trainer writes beside every `model.pt` (trainer.py:1010-1042), which holds the same
    configuration as plain YAML."""
    folder = Path(folder)
    assert folder.exists(), f"Model directory {folder} does not exist."
    steps = _step_dirs(folder)
    if step_nr is None:
        if not steps:
            raise ValueError(f"In {folder / 'checkpoints'}, no step number with model checkpoints was detected!")
        step_nr = steps[-1][0]
    step_dir = folder / "checkpoints" / f"step_{step_nr}"
    card_path = step_dir / "model_card.yaml"
    if not card_path.exists():
        raise ValueError(f"{step_dir} holds no model_card.yaml: the model configuration of this checkpoint is only in the "
                         "trainer's pickled metadata, which is not read here")
    card = _read_yaml(card_path)
    card = {**card, "name": card_name, "checkpoint": "file://" + (step_dir / "model.pt").absolute().as_posix()}
    register_card(card)
    return card


def load_model_with_overrides(model_dir: Union[str, Path], step: Optional[int] = None, model_type: Optional[str] = None,
                              device=None, dtype=None, model_filename: str = "model.pt"):
    """A model from a training directory (reference :72-125): through the `model_card.yaml` beside the checkpoint, or,
    for old checkpoints, through `config_logs/all_config.yaml` (then `model_type` must be given)."""
    model_dir = Path(model_dir)
    checkpoint_path = (model_dir / f"checkpoints/step_{step}" / model_filename) if step is not None else model_dir / model_filename
    card_path = checkpoint_path.parent / "model_card.yaml"
    if card_path.exists():
        try:
            return load_model_from_card(card_path.as_posix(), device=device, dtype=dtype)
        except Exception as exc:  # noqa: BLE001  (the reference falls back to the global config the same way)
            import logging
            logging.getLogger(__name__).warning(f"Model card {card_path} exists but is not valid ({exc}). Try global config instead.")
    config_path = model_dir / "config_logs/all_config.yaml"
    if config_path.exists():
        assert model_type, f"Need explicit model_type for checkpoint {checkpoint_path}"
        with open(config_path, "r", encoding="utf-8") as f:
            model_config = yaml.safe_load(f)["trainer"]["model_config_or_name"]
        card = create_model_card(checkpoint_path=checkpoint_path, model_config=model_config, model_type=model_type,
                                 model_arch=f"toy_{model_type}")
        from lcm.utils.model_type_registry import lcm_model_type_registry
        return lcm_model_type_registry.get_model_loader(model_type)(card, device=device, dtype=dtype)
    raise ValueError(f"{model_dir} is not a valid model directory")


def load_model_from_card(model_name_or_path: Union[str, Path, Dict[str, Any]], device=None, dtype=None):
    """`model_name_or_path` is, in this order (reference :178-223): a card (dict), the name of a card, the path of a
    card's YAML file (or of a directory holding `model_card.yaml`), a training directory (its last checkpoint), or a
    checkpoint `.../checkpoints/step_<n>/model.pt` inside one."""
    from lcm.utils.model_type_registry import lcm_model_type_registry

    try:
        card = resolve_card(model_name_or_path)
    except ValueError:
        path = Path(str(model_name_or_path))
        if (path / "checkpoints").exists():
            card = create_model_card_from_training_folder(path, card_name="temporary_card")
        elif path.suffix == ".pt" and path.parent.name.startswith("step_") and path.parent.parent.name == "checkpoints":
            card = create_model_card_from_training_folder(path.parent.parent.parent, card_name="temporary_card",
                                                          step_nr=int(path.parent.name[5:]))
        else:
            raise
    loader = lcm_model_type_registry.get_model_loader(card["model_family"])
    return loader(card, device=device, dtype=dtype)
