from lcm.models.sonar_normalizer.builder import (  # noqa: F401
    SonarNormalizer,
    SonarNormalizerConfig,
    create_sonar_normalizer,
    sonar_normalizer_archs,
)
from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model  # noqa: F401
