from lcm.inference.lcm.generator import LCMGenerator, LCMGeneratorOptions  # noqa: F401
from lcm.inference.lcm.scorer import LCMScorer  # noqa: F401
