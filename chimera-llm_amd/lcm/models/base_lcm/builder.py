"""Base (MSE) LCM whose arithmetic runs in liblcm_hip.so: frontend -> causal decoder -> postnet.

Public surface kept from the reference's lcm/models/base_lcm/builder.py: `BaseLCModelConfig` (:37-57), the
`lcm_archs` registry (:60-61), `BaseLCModel.{forward, predict_next_sentence}` (:94-167) and
`create_base_lcm_model` (:271-285); `LCMFrontendConfig` follows base_lcm/frontend.py:25-45.  Parameters are plain
device tensors named like the reference's state dict (`frontend.pre_linear.*`, `lcm.layers.{i}.*`,
`lcm.layer_norm.*`, `postnet.fc.*`, optional `frontend.sonar_normalizer.{center,scale}`).
"""

import ctypes as C
import logging
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from lcm import _hip
from lcm.nn.weight_norm import fold_weight_norm
from lcm.datasets.batch import EmbeddingsBatch
from lcm.models.abstract_lcm import AbstractLCModel, AbstractLCModelBuilder, AbstractLCModelConfig
from lcm.models.sonar_normalizer.builder import SonarNormalizer, SonarNormalizerConfig
from lcm.models.two_tower_diffusion_lcm.builder import (
    TIMESTEP_FREQ_DIM,
    DeviceSession,
    _ArchRegistry,
    _check_supported,
    extra_layer_spec,
    final_norm_absent,
    normalizer_config_for,
    normalizer_spec,
    prelu_spec,
    tower_desc,
    load_named_normalizer_statistics,
)
from lcm.nn.incremental_state import LCMIncrementalStateBag
from lcm.nn.transformer_config import ProjectionConfig, TransformerConfig

logger = logging.getLogger(__name__)

BASE_LCM_MODEL_TYPE = "base_lcm"


@dataclass
class LCMFrontendConfig:
    dropout_p: float = 0.0
    pre_linear_bias: bool = True
    pre_linear_init_fn: str = "kaiming_uniform"
    scale_embeddings: bool = False       # multiply the inputs by sqrt(model_dim) before the pre-linear
    weight_normalization: bool = False
    embedding_std: float = 0.006         # SONAR_STD of lcm/nn/initialization.py


@dataclass
class BaseLCModelConfig(AbstractLCModelConfig):
    model_type: str = BASE_LCM_MODEL_TYPE
    max_seq_len: int = 2048
    model_dim: int = 1024
    model_output_dim: Optional[int] = None
    frontend: LCMFrontendConfig = field(default_factory=LCMFrontendConfig)
    lcm: TransformerConfig = field(default_factory=TransformerConfig)
    postnet: ProjectionConfig = field(default_factory=ProjectionConfig)


lcm_archs = _ArchRegistry()
lcm_arch = lcm_archs.decorator


def parameter_spec(config: BaseLCModelConfig,
                   normalizer: Optional[SonarNormalizerConfig] = None) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) of every parameter/buffer, named like the reference's state dict."""
    d, D, tc = config.model_dim, config.sonar_embed_dim, config.lcm
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    if config.sonar_normalizer_name is not None:
        spec += normalizer_spec("frontend.sonar_normalizer", normalizer or SonarNormalizerConfig(dim=D))
    spec.append(("frontend.pre_linear.weight", (d, D), "matrix"))
    if config.frontend.pre_linear_bias:
        spec.append(("frontend.pre_linear.bias", (d,), "zeros"))
    if tc.pos_embedding_style == "learned":
        spec.append(("frontend.pos_encoder.weight", (config.max_seq_len, d), "pos"))
    F = tc.allocated_ffn_dim()
    ln_weight = tc.layer_normalization_style != "unit"
    ln_bias = tc.layer_normalization_style == "standard"

    def norm(n: str) -> None:
        if ln_weight:
            spec.append((f"{n}.weight", (d,), "ones"))
        if ln_bias:
            spec.append((f"{n}.bias", (d,), "zeros"))

    for i in range(tc.num_layers):
        p = f"lcm.layers.{i}"
        norm(f"{p}.self_attn_layer_norm")
        norm(f"{p}.ffn_layer_norm")
        for proj in ("q_proj", "k_proj", "v_proj"):
            spec.append((f"{p}.self_attn.{proj}.weight", (d, d), "matrix"))
            spec.append((f"{p}.self_attn.{proj}.bias", (d,), "zeros"))
        spec.append((f"{p}.self_attn.output_proj.weight", (d, d), "matrix"))
        if tc.mha_output_proj_bias:
            spec.append((f"{p}.self_attn.output_proj.bias", (d,), "zeros"))
        if tc.enable_qk_layernorm:       # lcm/nn/transformer/attention.py:131-141
            norm(f"{p}.self_attn.q_layer_norm")
            norm(f"{p}.self_attn.k_layer_norm")
        if tc.use_swiglu:
            spec.append((f"{p}.ffn.gate_proj.weight", (F, d), "matrix"))
            spec.append((f"{p}.ffn.gate_proj.bias", (F,), "zeros"))
        spec.append((f"{p}.ffn.inner_proj.weight", (F, d), "matrix"))
        spec.append((f"{p}.ffn.inner_proj.bias", (F,), "zeros"))
        spec.append((f"{p}.ffn.output_proj.weight", (d, F), "matrix"))
        spec.append((f"{p}.ffn.output_proj.bias", (d,), "zeros"))
        prelu_spec(spec, p, tc)
        extra_layer_spec(spec, p, tc, d, norm)
    if not final_norm_absent(tc):
        norm("lcm.layer_norm")
    spec.append(("postnet.fc.weight", (D, d), "matrix"))
    if config.postnet.linear_bias:
        spec.append(("postnet.fc.bias", (D,), "zeros"))
    return spec


class _Frontend:
    """What callers read from `model.frontend` (reference: base_lcm/frontend.py:96)."""

    def __init__(self, sonar_normalizer: Optional[SonarNormalizer]) -> None:
        self.sonar_normalizer = sonar_normalizer


class BaseLCModel(AbstractLCModel):
    config: BaseLCModelConfig

    def __init__(self, config: BaseLCModelConfig, *, device=None, dtype: Optional[torch.dtype] = None,
                 seed: Optional[int] = None) -> None:
        super().__init__(config)
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("BaseLCModel runs only on a GPU (HIP path); there is no CPU fallback")
        dtype = dtype or torch.float32
        _hip.dtype_code(dtype)
        _check_supported("lcm", config.lcm)
        if config.model_output_dim not in (None, config.sonar_embed_dim):
            raise NotImplementedError("model_output_dim other than the SONAR dimension is not implemented on the HIP path")
        if config.postnet.activation_name is not None and (config.postnet.activation_name not in _hip.ACT
                                                           or config.postnet.activation_name == "prelu"):
            # "prelu" carries a learned slope; anything else is unknown to the reference too (initialization.py:121-151)
            raise NotImplementedError(f"postnet activation {config.postnet.activation_name} is not implemented on the HIP path")
        self.model_dim = config.model_dim
        self.sonar_embed_dim = config.sonar_embed_dim
        ncfg = normalizer_config_for(config.sonar_normalizer_name, config.sonar_embed_dim)
        self._spec = parameter_spec(config, ncfg)
        self._params: Dict[str, Tensor] = {
            name: torch.empty(shape, dtype=dtype, device=device) for name, shape, _ in self._spec}
        self._handle = None
        self.init_parameters(seed if seed is not None else 0)
        normalizer = None
        if config.sonar_normalizer_name is not None:
            normalizer = SonarNormalizer(ncfg, device=device, dtype=dtype)
            normalizer.center = self._params["frontend.sonar_normalizer.center"]
            normalizer.scale = self._params["frontend.sonar_normalizer.scale"]
            if ncfg.clip_proba is not None:
                normalizer.clip_min = self._params["frontend.sonar_normalizer.clip_min"]
                normalizer.clip_max = self._params["frontend.sonar_normalizer.clip_max"]
            load_named_normalizer_statistics(normalizer, config.sonar_normalizer_name)
        self.frontend = _Frontend(normalizer)
        self.sonar_normalizer = normalizer
        self._handle = C.c_void_p()
        self._bind()
        self._session_pool: List[DeviceSession] = []
        self._session_lock = threading.Lock()      # generators on several host threads share one model
        self.session_flags = 0     # _hip.SESSION_* bits given to every session this model creates (tests, profiling)

    # ---- parameters ---------------------------------------------------------------------------------
    def init_parameters(self, seed: int = 0, matrix_std: float = 0.02) -> None:
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, shape, kind in self._spec:
            t = self._params[name]
            if kind == "zeros":
                t.zero_()
            elif kind == "ones":
                t.fill_(1.0)
            elif kind.startswith("const:"):
                t.fill_(float(kind[6:]))
            else:
                std = matrix_std if kind == "matrix" else self.config.frontend.embedding_std
                t.copy_(torch.randn(shape, generator=g, device=t.device, dtype=torch.float32) * std)

    def state_dict(self) -> Dict[str, Tensor]:
        return self._params

    def load_state_dict(self, state_dict: Dict[str, Tensor], strict: bool = True) -> None:
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        sd = fold_weight_norm(sd)      # weight-normalized projections of a checkpoint become plain matrices
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        for k, dst in self._params.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(dst.shape):
                    raise RuntimeError(f"load_state_dict: shape mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(dst.shape)}")
                dst.copy_(sd[k])

    def num_parameters(self) -> int:
        return sum(t.numel() for n, t in self._params.items() if "sonar_normalizer." not in n)

    def model_desc(self) -> "_hip.ModelDesc":
        """The lcmhip_model_desc this model was created from."""
        c, tc = self.config, self.config.lcm
        tower = tower_desc(tc)
        none = _hip.TowerDesc(num_layers=0)      # no denoiser tower: selects the base LCM in the library
        return _hip.ModelDesc(sonar_dim=c.sonar_embed_dim, model_dim=c.model_dim, max_seq_len=c.max_seq_len,
                              timestep_freq_dim=TIMESTEP_FREQ_DIM, dtype=_hip.dtype_code(self.dtype), norm_eps=1e-5,
                              context=tower, denoiser=none,
                              embed_scale=float(c.model_dim ** 0.5) if c.frontend.scale_embeddings else 1.0,
                              postnet_activation=0 if c.postnet.activation_name is None else 1 + _hip.ACT[c.postnet.activation_name])

    def bound_tensors(self) -> Dict[str, Tensor]:
        """Every tensor bound into the library's weight table, by name (the state dict: a base LCM has no derived ones)."""
        return dict(self._params)

    def _bind(self) -> None:
        desc = self.model_desc()
        _hip.check(_hip.lib.lcmhip_model_create(C.byref(desc), C.byref(self._handle)))
        code = _hip.dtype_code(self.dtype)
        for name, t in self._params.items():
            shape = (C.c_int64 * t.dim())(*t.shape)
            _hip.check(_hip.lib.lcmhip_model_bind_weight(self._handle, name.encode(), _hip.ptr(t), code, t.dim(), shape))
        _hip.check(_hip.lib.lcmhip_model_finalize(self._handle))

    def __del__(self) -> None:
        pool = getattr(self, "_session_pool", None)
        if pool:
            pool.clear()
        h, self._handle = getattr(self, "_handle", None), None
        if h and _hip is not None and getattr(_hip, "lib", None) is not None:
            _hip.lib.lcmhip_model_destroy(h)

    # ---- sessions -----------------------------------------------------------------------------------
    def acquire_session(self, max_batch: int, max_ctx: int) -> DeviceSession:
        sess = None
        with self._session_lock:
            for i, s in enumerate(self._session_pool):
                if s.dims[0] >= max_batch and s.dims[1] >= max_ctx and s.flags == int(self.session_flags):
                    sess = self._session_pool.pop(i)
                    break
            else:
                self._session_pool.clear()
            s = None                         # the loop variable would keep the dropped session's workspace alive below
        if sess is None:
            return DeviceSession(self, max_batch, max_ctx, 1)
        sess.unpark()
        return sess

    def release_session(self, sess: Optional[DeviceSession]) -> None:
        if sess is None:
            return
        sess.park()
        with self._session_lock:
            if not self._session_pool:
                self._session_pool.append(sess)

    def _session_for(self, bag: LCMIncrementalStateBag, batch_size: int) -> DeviceSession:
        if bag.session is None:
            bag.session = self.acquire_session(batch_size, bag.max_num_steps)
            bag.session.reset(batch_size)
        return bag.session

    # ---- reference API ------------------------------------------------------------------------------
    @torch.inference_mode()
    def _run(self, seqs: Tensor, state_bag: Optional[LCMIncrementalStateBag], *, normalized: bool,
             noise: Optional[Tensor], temperature: float, want_output: bool = True) -> Optional[Tensor]:
        B, S, _ = seqs.shape
        if state_bag is None:          # no KV cache kept: a throw-away session sized for this call
            sess = self.acquire_session(B, S)
            sess.reset(B)
        else:
            sess = self._session_for(state_bag, B)
        try:
            return sess.base_forward(seqs, normalized=normalized, noise=noise, temperature=temperature,
                                     all_positions=True, want_output=want_output)
        finally:
            if state_bag is None:
                self.release_session(sess)

    def forward(self, batch: EmbeddingsBatch, state_bag: Optional[LCMIncrementalStateBag] = None, **kwargs) -> EmbeddingsBatch:
        """Frontend, causal decoder and postnet (reference :94-124): no normalizer on either side, so inputs in the
        normalizer's space give predictions in the normalizer's space."""
        if batch.padding_mask is not None:
            raise ValueError("the HIP decoder takes unpadded positions (the generators never pad them)")
        out = self._run(batch.seqs, state_bag, normalized=True, noise=None, temperature=0.0)
        return EmbeddingsBatch(out, None)

    __call__ = forward

    def predict_next_sentence(self, batch: EmbeddingsBatch, sample: bool = False, temperature: float = 1.0,
                              state_bag: Optional[LCMIncrementalStateBag] = None, noise: Optional[Tensor] = None,
                              **kwargs) -> EmbeddingsBatch:
        """Raw SONAR embeddings (B, s, D) -> predictions (B, s, D) fp32 for every position (reference :126-167).
        With `sample` and temperature > 0, N(0, temperature^2) noise is added to the normalized prediction; `noise`
        (B, s, D) supplies the N(0, 1) draws, otherwise they come from torch.randn on the device."""
        if batch.padding_mask is not None:
            raise ValueError("the HIP decoder takes unpadded positions (the generators never pad them)")
        x = batch.seqs
        nz = None
        if sample and temperature > 0:
            nz = noise.reshape(x.shape) if noise is not None else torch.randn(x.shape, device=x.device)
        out = self._run(x, state_bag, normalized=False, noise=nz, temperature=float(temperature) if nz is not None else 0.0)
        return EmbeddingsBatch(out, None)

    def prefill(self, batch: EmbeddingsBatch, state_bag: LCMIncrementalStateBag) -> None:
        """Fills the KV cache with raw prompt positions without computing outputs (what the generator's prefill
        needs: lcm/inference/lcm/generator.py:226-246 discards the prediction)."""
        self._run(batch.seqs, state_bag, normalized=False, noise=None, temperature=0.0, want_output=False)


class BaseLCModelBuilder(AbstractLCModelBuilder):
    """The reference's builder class (base_lcm/builder.py:170-272) as far as it applies here: `build_model` and the
    inherited `build_sonar_normalizer`; its other `build_*` methods make torch modules, which this implementation does
    not have."""

    config: BaseLCModelConfig

    def build_model(self) -> "BaseLCModel":
        return BaseLCModel(self.config, device=self.device, dtype=self.dtype)


def create_base_lcm_model(config: BaseLCModelConfig, *, device=None, dtype: Optional[torch.dtype] = None,
                          seed: Optional[int] = None) -> BaseLCModel:
    return BaseLCModel(config, device=device, dtype=dtype, seed=seed)
