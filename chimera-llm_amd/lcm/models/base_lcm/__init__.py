# Importing the archs module registers the named architectures.
from lcm.models.base_lcm import archs as _archs  # noqa: F401
from lcm.models.base_lcm.builder import (  # noqa: F401
    BASE_LCM_MODEL_TYPE,
    BaseLCModel,
    BaseLCModelBuilder,
    BaseLCModelConfig,
    LCMFrontendConfig,
    ProjectionConfig,
    TransformerConfig,
    create_base_lcm_model,
    lcm_archs,
)
from lcm.models.base_lcm.loader import load_base_lcm_model  # noqa: F401
