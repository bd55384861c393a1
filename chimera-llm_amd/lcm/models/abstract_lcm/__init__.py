from lcm.models.abstract_lcm.builder import AbstractLCModel, AbstractLCModelConfig  # noqa: F401
