"""`LCMConfig` / `LCMPredictor`: the evaluation-side caller of the base generator (reference
lcm/evaluation/predictors/lcm.py:33-269).

The embedding half is the reference's: model from a card, generator from the config (which IS the generator's option
set), prompts cut / batched / padded, the best hypothesis without its prompt per document.  The text half needs the
SONAR text encoder and decoder, which this package does not contain: a caller who has them passes objects with the
pipelines' `predict` method as `text_encoder=` / `text_decoder=` (and a sentence splitter as `split_text=`); without a
decoder the predictions carry embeddings only (`text` is None), and text prompts without an encoder are refused."""

import logging
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, Iterable, List, Optional, Sequence

import torch

from lcm.datasets.configs import SonarDecoderConfig, SonarEncoderConfig
from lcm.evaluation.api import EOSConfig, Prediction, PredictorConfig
from lcm.inference.lcm import LCMGenerator, LCMGeneratorOptions
from lcm.inference.predict import generate_continuations
from lcm.utils.card_utils import load_model_from_card


logger = logging.getLogger(__name__)


def torch_type(dtype: Any) -> torch.dtype:
    """"torch.bfloat16" / "bfloat16" / torch.bfloat16 / None -> a dtype the HIP path computes in (the reference's
    evaluation default, float16, is not one: such runs are served in bfloat16)."""
    if dtype is None:
        return torch.bfloat16
    if isinstance(dtype, str):
        dtype = getattr(torch, dtype.replace("torch.", ""))
    if dtype == torch.float16:
        logger.warning("float16 is not a dtype of the HIP path: the model is loaded and run in bfloat16")
        return torch.bfloat16
    return dtype


@dataclass(unsafe_hash=True)
class LCMConfig(LCMGeneratorOptions, PredictorConfig):
    model_card: str = ""
    decoder_config: SonarDecoderConfig = field(default_factory=SonarDecoderConfig)
    encoder_config: SonarEncoderConfig = field(default_factory=SonarEncoderConfig)
    generator_batch_size: Optional[int] = None      # None: the whole task batch at once

    @classmethod
    def predictor_class(cls):
        return LCMPredictor


class LCMPredictor:
    def __init__(self, config: LCMConfig, **kwargs: Any) -> None:
        self.config = config
        # the run's temperature wins over the configured one unless it is 0 (reference lcm.py:74-83)
        if kwargs.get("temperature", 0) > 0:
            self.config.lcm_temperature = kwargs["temperature"]
        for k, v in kwargs.items():
            if hasattr(self.config, k) and v:
                setattr(self.config, k, v)
        self.device = torch.device(kwargs.get("device") or "cuda:0")
        self.dtype = torch_type(kwargs.get("dtype"))
        self.text_encoder = kwargs.get("text_encoder")
        self.text_decoder = kwargs.get("text_decoder")
        self.split_text: Optional[Callable[[str], Iterable[str]]] = kwargs.get("split_text")
        model = kwargs.get("model") or load_model_from_card(self.config.model_card, device=self.device, dtype=self.dtype)
        self.eos_vec = self.get_eos(kwargs.get("eos_config"))
        self.build_generator(model)

    def get_eos(self, eos_config: Optional[EOSConfig]) -> Optional[torch.Tensor]:
        if not eos_config:
            return None
        if eos_config.ckpt and Path(eos_config.ckpt).exists():
            eos_vec = torch.load(eos_config.ckpt, map_location=self.device, weights_only=True)
        elif eos_config.text:
            eos_vec = self._encode([eos_config.text]).squeeze()
        else:
            return None
        return eos_vec.to(device=self.device, dtype=self.dtype)

    def build_generator(self, model) -> None:
        self.generator = LCMGenerator(model, self.config, eos_vec=self.eos_vec)

    @staticmethod
    def from_config(config: LCMConfig, **kwargs: Any) -> "LCMPredictor":
        return LCMPredictor(config=config, **kwargs)

    def _encode(self, sentences: Sequence[str]) -> torch.Tensor:
        if self.text_encoder is None:
            raise RuntimeError("text prompts need a SONAR text encoder: pass text_encoder= (an object with the "
                               "TextToEmbeddingModelPipeline.predict method), or hand over encoded prompts")
        return self.text_encoder.predict(list(sentences), source_lang=self.config.encoder_config.lang).to(self.device)

    def format_predictions(self, continuations: List[torch.Tensor], step_scores=None) -> List[Prediction]:
        """All documents' sentences go through the text decoder in one call, as in the reference (lcm.py:126-176)."""
        texts: List[Optional[List[str]]] = [None] * len(continuations)
        if self.text_decoder is not None and continuations:
            dc = self.config.decoder_config
            flat = self.text_decoder.predict(torch.cat(continuations), target_lang=dc.lang,
                                             max_seq_len=dc.max_tokens_in_sentence, temperature=dc.temperature)
            start = 0
            for i, e in enumerate(continuations):
                texts[i] = list(flat[start:start + e.shape[0]])
                start += e.shape[0]
        return [Prediction(text=t, embed=e, logprobs=None if step_scores is None else step_scores[i])
                for i, (t, e) in enumerate(zip(texts, continuations))]

    @torch.inference_mode()
    def __call__(self, prompts, max_prompt_len: Optional[int] = None, min_gen_len: Optional[int] = None,
                 max_gen_len: Optional[int] = None, temperature: float = 0.0, disable_cache: bool = False,
                 greedy: bool = True, top_p: float = 0.0, top_k: int = 0, echo: bool = True,
                 return_logprobs: bool = False, show_progress: bool = False, **kwargs: Any) -> List[Prediction]:
        assert greedy, "Non-greedy generation for continuous space not yet supported"
        assert not return_logprobs, "the HIP generators keep no per-step scores (the reference's are dummies)"
        if max_gen_len is None:
            max_gen_len = self.config.max_seq_len
        assert isinstance(prompts, (list, tuple)) and len(prompts) > 0, f"Expect sequence of prompts, get {type(prompts)}"
        if isinstance(prompts[0], str):
            split = self.split_text or (lambda text: [text])
            seqs = [self._encode(list(split(str(p)))) for p in prompts]
        else:
            assert all(isinstance(p, torch.Tensor) for p in prompts), \
                f"Expect sonarized prompts in the form or List[torch.Tensor], get {type(prompts[0])}"
            seqs = [p.to(self.device) for p in prompts]
        continuations = generate_continuations(
            self.generator, seqs, max_gen_len=max_gen_len, min_gen_len=min_gen_len,
            max_gen_len_ratio=kwargs.get("max_gen_len_ratio"), max_prompt_len=max_prompt_len,
            generator_batch_size=self.config.generator_batch_size, temperature=temperature, disable_cache=disable_cache)
        return self.format_predictions(continuations)
