"""model_family -> loader (reference: lcm/utils/model_type_registry.py:20-72)."""

from typing import Callable, Dict


class ModelTypeRegistry:
    def __init__(self) -> None:
        self._loaders: Dict[str, Callable] = {}

    def register(self, model_type: str, loader: Callable) -> None:
        self._loaders[model_type] = loader

    def get_model_loader(self, model_type: str) -> Callable:
        if model_type not in self._loaders:
            raise ValueError(f"unknown model_family '{model_type}' (known: {sorted(self._loaders)})")
        return self._loaders[model_type]


lcm_model_type_registry = ModelTypeRegistry()


def _register_defaults() -> None:
    from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model
    from lcm.models.two_tower_diffusion_lcm.loader import load_two_tower_diffusion_lcm_model

    from lcm.models.base_lcm.loader import load_base_lcm_model

    lcm_model_type_registry.register("two_tower_diffusion_lcm", load_two_tower_diffusion_lcm_model)
    lcm_model_type_registry.register("base_lcm", load_base_lcm_model)
    lcm_model_type_registry.register("sonar_normalizer", load_sonar_normalizer_model)


_register_defaults()
