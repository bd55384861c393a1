"""The two SONAR pipeline configurations the evaluation predictors carry (reference lcm/datasets/configs.py:60-88).
They only name models: the SONAR text encoder / decoder themselves are not part of this package."""

from dataclasses import dataclass


@dataclass(frozen=True)
class SonarDecoderConfig:
    tokenizer: str = "text_sonar_basic_decoder"
    decoder: str = "text_sonar_basic_decoder"
    lang: str = "eng_Latn"
    max_tokens_in_sentence: int = 256
    temperature: float = 1.0


@dataclass(frozen=True)
class SonarEncoderConfig:
    tokenizer: str = "text_sonar_basic_encoder"
    encoder: str = "text_sonar_basic_encoder"
    lang: str = "eng_Latn"
