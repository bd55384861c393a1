"""model_family -> (config loader, model factory, model loader) (reference: lcm/utils/model_type_registry.py:9-72)."""

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Union


@dataclass
class ModelTypeConfig:
    """The functions that belong to one model type."""

    model_type: str
    config_loader: Callable
    model_factory: Callable
    model_loader: Callable


class ModelTypeRegistry:
    def __init__(self) -> None:
        self._configs: Dict[str, ModelTypeConfig] = {}

    def register(self, model_type_config: Union[ModelTypeConfig, str], loader: Optional[Callable] = None) -> None:
        """`register(ModelTypeConfig(...))` as in the reference; `register(name, loader)` is the short form for a type
        that only has a loader."""
        if not isinstance(model_type_config, ModelTypeConfig):
            assert loader is not None, "register(name, loader) needs the loader"
            model_type_config = ModelTypeConfig(str(model_type_config), config_loader=None, model_factory=None, model_loader=loader)
        model_type = model_type_config.model_type
        assert model_type, "To register a model type, the model_type parameter should be non-empty."
        if model_type in self._configs:
            raise ValueError(f"`model_type` must be a unique model type name, but '{model_type}' is already registered.")
        self._configs[model_type] = model_type_config

    def get_config(self, model_type: str) -> ModelTypeConfig:
        try:
            return self._configs[model_type]
        except KeyError:
            raise ValueError(f"The registry of model types does not contain a model type named '{model_type}'.")

    def get_model_loader(self, model_type: str) -> Callable:
        return self.get_config(model_type).model_loader


lcm_model_type_registry = ModelTypeRegistry()


def _register_defaults() -> None:
    from lcm.models.base_lcm.builder import create_base_lcm_model
    from lcm.models.base_lcm.loader import load_base_lcm_config, load_base_lcm_model
    from lcm.models.sonar_normalizer.builder import create_sonar_normalizer
    from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_config, load_sonar_normalizer_model
    from lcm.models.two_tower_diffusion_lcm.builder import create_two_tower_diffusion_lcm_model
    from lcm.models.two_tower_diffusion_lcm.loader import load_two_tower_diffusion_lcm_config, load_two_tower_diffusion_lcm_model

    for name, cfg, factory, loader in (
            ("two_tower_diffusion_lcm", load_two_tower_diffusion_lcm_config, create_two_tower_diffusion_lcm_model, load_two_tower_diffusion_lcm_model),
            ("base_lcm", load_base_lcm_config, create_base_lcm_model, load_base_lcm_model),
            ("sonar_normalizer", load_sonar_normalizer_config, create_sonar_normalizer, load_sonar_normalizer_model)):
        lcm_model_type_registry.register(ModelTypeConfig(name, config_loader=cfg, model_factory=factory, model_loader=loader))


_register_defaults()
