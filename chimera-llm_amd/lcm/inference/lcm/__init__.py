from lcm.inference.lcm.generator import LCMGenerator, LCMGeneratorOptions  # noqa: F401
