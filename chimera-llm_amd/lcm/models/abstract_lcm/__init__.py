from lcm.models.abstract_lcm.builder import AbstractLCModel, AbstractLCModelBuilder, AbstractLCModelConfig  # noqa: F401
