"""SONAR embedding normalizer on the HIP path.

`normalize` / `denormalize` follow the reference's lcm/models/sonar_normalizer/builder.py:121-137
(centre/scale, optional quantile clipping) and run in the coalesced `lcmhip_normalize` /
`lcmhip_denormalize` kernels; the packed real FFT of the "*_fft" archs (`FFTInterface`, builder.py:58-84)
runs in `lcmhip_rfft_pack`.  `fit` (:139-193) is the reference's host-side scikit-learn pass.
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

    @torch.no_grad()
    def fit(self, embeddings: Tensor) -> None:
        """Fits center / scale (and the clipping quantiles) to `embeddings` (N, dim), as the reference does
        (builder.py:139-193): scikit-learn's RobustScaler ("robust", "gaussian_robust": median and the spread between the
        configured quantiles, which scikit-learn reads as PERCENT, so the defaults 0.25 / 0.75 are the 0.25th and 0.75th
        percentile) or StandardScaler ("standard"), on the host in fp32; the clipping bounds are quantiles of the scaled
        data.  A host-side, one-off statistics pass: no kernel of this library is involved except the packed FFT of the
        "*_fft" archs, which needs the embeddings on the GPU."""
        method = self.config.normalization_method
        if method in ("robust", "gaussian_robust"):
            from sklearn.preprocessing import RobustScaler
            scaler = RobustScaler(unit_variance=method == "gaussian_robust",
                                  quantile_range=(self.config.quantile_min, self.config.quantile_max))
        elif method == "standard":
            from sklearn.preprocessing import StandardScaler
            scaler = StandardScaler()
        else:
            raise ValueError(f"Unrecognizable method {method} for scaling input features")
        assert embeddings.dim() == 2 and embeddings.shape[-1] == self.config.dim
        if self.config.with_fft:
            embeddings = self.fft_transform(embeddings)
        scaled = scaler.fit_transform(embeddings.cpu().float().numpy())
        center = scaler.mean_ if method == "standard" else scaler.center_
        self.center.copy_(torch.tensor(center, dtype=self.center.dtype))
        self.scale.copy_(torch.tensor(scaler.scale_, dtype=self.scale.dtype))
        if self.config.clip_proba is not None:
            t = torch.tensor(scaled)
            self.clip_min.copy_(torch.quantile(t, self.config.clip_proba, dim=0).to(self.clip_min.dtype))
            self.clip_max.copy_(torch.quantile(t, 1 - self.config.clip_proba, dim=0).to(self.clip_max.dtype))


def create_sonar_normalizer(config: SonarNormalizerConfig, *, device=None, dtype=None) -> SonarNormalizer:
    return SonarNormalizer(config, device=device, dtype=dtype)
