"""SONAR embedding normalizer on the HIP path.

`normalize` / `denormalize` follow the reference's lcm/models/sonar_normalizer/builder.py:121-137
(centre/scale, optional quantile clipping) and run in the coalesced `lcmhip_normalize` /
`lcmhip_denormalize` kernels; the packed real FFT of the "*_fft" archs (`FFTInterface`, builder.py:58-84)
runs in `lcmhip_rfft_pack`.  `fit` (:139-193, scikit-learn on the host) is not part of the generation
path and raises.
"""

from dataclasses import dataclass
from typing import Dict, Literal, Optional

import torch
from torch import Tensor

from lcm import _hip


@dataclass
class SonarNormalizerConfig:
    dim: int = 1024
    clip_proba: Optional[float] = None
    with_fft: bool = False
    quantile_min: float = 0.25
    quantile_max: float = 0.75
    normalization_method: Literal["standard", "robust", "gaussian_robust"] = "gaussian_robust"


sonar_normalizer_archs: Dict[str, SonarNormalizerConfig] = {
    "base": SonarNormalizerConfig(),
    "base_page4k": SonarNormalizerConfig(dim=4 * 1024),
    "clipping": SonarNormalizerConfig(clip_proba=1e-4),
    "base_fft": SonarNormalizerConfig(with_fft=True),
    "clipping_fft": SonarNormalizerConfig(clip_proba=1e-4, with_fft=True),
}


class FFTInterface:
    """The reference's mix-in (builder.py:58-84): a real FFT over the embedding axis packed into as many real numbers,
    [Re X_0 .. Re X_{n/2}, Im X_1 .. Im X_{n/2-1}], and its inverse (GPU tensors, widths that are a power of two)."""

    @staticmethod
    def fft_transform(embeddings: Tensor) -> Tensor:
        return _hip.rfft_pack(embeddings, inverse=False).to(embeddings.dtype)

    @staticmethod
    def fft_inverse_transform(embeddings: Tensor) -> Tensor:
        return _hip.rfft_pack(embeddings, inverse=True).to(embeddings.dtype)


class SonarNormalizer(FFTInterface):
    def __init__(self, config: SonarNormalizerConfig, device=None, dtype: Optional[torch.dtype] = None) -> None:
        self.config = config
        dtype = dtype or torch.float32
        self.center = torch.zeros(config.dim, dtype=dtype, device=device)
        self.scale = torch.ones(config.dim, dtype=dtype, device=device)
        self.clip_min = self.clip_max = None
        if config.clip_proba is not None:
            self.clip_min = torch.ones(config.dim, dtype=dtype, device=device)
            self.clip_max = torch.ones(config.dim, dtype=dtype, device=device)

    def state_dict(self) -> Dict[str, Tensor]:
        sd = {"center": self.center, "scale": self.scale}
        if self.clip_min is not None:
            sd.update(clip_min=self.clip_min, clip_max=self.clip_max)
        return sd

    def load_state_dict(self, sd: Dict[str, Tensor]) -> None:
        for k, dst in self.state_dict().items():
            dst.copy_(sd[k].to(dst.dtype))

    def _run(self, x: Tensor, fn) -> Tensor:
        _hip.require_gpu(x, "embeddings")
        dt = self.center.dtype
        xin = x.to(dt).contiguous()
        y = torch.empty_like(xin)
        dim = xin.shape[-1]
        if dim != self.config.dim:
            raise ValueError(f"expected embeddings of width {self.config.dim}, got {dim}")
        _hip.check(fn(_hip.ptr(xin), _hip.ptr(self.center), _hip.ptr(self.scale), _hip.ptr(self.clip_min),
                      _hip.ptr(self.clip_max), _hip.ptr(y), xin.numel() // dim, dim, _hip.dtype_code(dt),
                      _hip.stream_ptr(x.device)))
        return y

    def normalize(self, embeddings: Tensor) -> Tensor:
        if self.config.with_fft:        # builder.py:121-128: the transform precedes center / scale / clip
            embeddings = _hip.rfft_pack(embeddings, inverse=False)
        return self._run(embeddings, _hip.lib.lcmhip_normalize)

    def denormalize(self, embeddings: Tensor) -> Tensor:
        y = self._run(embeddings, _hip.lib.lcmhip_denormalize)
        if self.config.with_fft:        # builder.py:130-137: and its inverse follows them
            y = _hip.rfft_pack(y, inverse=True).to(y.dtype)
        return y

    def fit(self, embeddings: Tensor):
        raise NotImplementedError("fitting a normalizer is outside the generation path")


def create_sonar_normalizer(config: SonarNormalizerConfig, *, device=None, dtype=None) -> SonarNormalizer:
    return SonarNormalizer(config, device=device, dtype=dtype)
