"""Two-tower diffusion LCM whose arithmetic runs in liblcm_hip.so.

Public surface kept from the reference's lcm/models/two_tower_diffusion_lcm/builder.py:
`TwoTowerDiffusionLCModelConfig` (:49-81), the `lcm_archs` registry (:84-85),
`TwoTowerDiffusionLCModel.{encode, prep_for_denoising, predict_next_sentence,
predict_next_sentence_with_cf_guidance}` (:161-191, 215-242, 262-461) and
`create_two_tower_diffusion_lcm_model` (:610-628).  Parameters are plain device tensors named
like the reference's state dict; a `DeviceSession` owns the KV caches and scratch of one batch.
"""

import ctypes as C
import os
import threading
import math
import logging
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from lcm import _hip
from lcm.nn.weight_norm import fold_weight_norm
from lcm.datasets.batch import EmbeddingsBatch
from lcm.models.abstract_lcm import AbstractLCModel, AbstractLCModelBuilder, AbstractLCModelConfig
from lcm.models.sonar_normalizer.builder import SonarNormalizer, SonarNormalizerConfig
from lcm.nn.incremental_state import LCMIncrementalStateBag
from lcm.nn.schedulers import DDIMScheduler, DDIMSchedulerConfig
from lcm.nn.transformer_config import DenoiserConfig, EncoderFrontendConfig, TransformerConfig

logger = logging.getLogger(__name__)

TWO_TOWER_DIFFUSION_LCM_MODEL_TYPE = "two_tower_diffusion_lcm"


@dataclass
class TwoTowerDiffusionLCModelConfig(AbstractLCModelConfig):
    model_type: str = TWO_TOWER_DIFFUSION_LCM_MODEL_TYPE
    max_seq_len: int = 2048
    model_dim: int = 1024
    frontend: EncoderFrontendConfig = field(default_factory=EncoderFrontendConfig)
    context_encoder: TransformerConfig = field(default_factory=TransformerConfig)
    noise_scheduler: DDIMSchedulerConfig = field(default_factory=DDIMSchedulerConfig)
    denoiser: DenoiserConfig = field(default_factory=DenoiserConfig)
    trained_with_cf_guidance: bool = False


class _ArchRegistry(dict):
    def decorator(self, name: str) -> Callable:
        def register(fn):
            self[name] = fn
            return fn
        return register

    def get_config(self, name: str) -> TwoTowerDiffusionLCModelConfig:
        if name not in self:
            raise KeyError(f"unknown two_tower_diffusion_lcm architecture '{name}' (known: {sorted(self)})")
        return self[name]()

    # the calls of fairseq2's ConfigRegistry that the reference's code and tests use on `lcm_archs`
    def get(self, name: str, *unused):  # type: ignore[override]
        """A fresh configuration of the named architecture (not the registered factory, as dict.get would return)."""
        return self.get_config(name)

    def names(self):
        return set(self.keys())


lcm_archs = _ArchRegistry()
lcm_arch = lcm_archs.decorator

_NORM_KIND = {"standard": _hip.NORM_LAYERNORM, "rms": _hip.NORM_RMS, "fp32": _hip.NORM_FP32, "unit": _hip.NORM_UNIT}
_POS_KIND = {"none": _hip.POS_NONE, "learned": _hip.POS_LEARNED, "rope": _hip.POS_ROPE, "sine": _hip.POS_SINE}
TIMESTEP_FREQ_DIM = 256


def _check_supported(name: str, tc: TransformerConfig, denoiser: bool = False) -> None:
    """Options of a stack the HIP path does not implement raise here, at construction.  The denoiser's layers are
    pre-norm in the reference too (lcm/nn/denoisers/lcm_denoiser.py:441-443 asserts against "post"); its factory never
    forwards `scale_residual` (lcm/nn/denoisers/factory.py:181-191) and its stack's final norm follows norm_order_style
    alone (:132-151), so those two settings are accepted for the denoiser and have no effect, as in the reference."""
    bad = []
    if tc.layer_normalization_style not in _NORM_KIND:
        bad.append(f"layer_normalization_style={tc.layer_normalization_style}")
    if denoiser:
        if tc.norm_order_style != "pre":
            bad.append(f"norm_order_style={tc.norm_order_style} (denoiser layers are pre-norm)")
        if tc.pos_embedding_style not in ("none", "learned", "sine"):
            bad.append(f"pos_embedding_style={tc.pos_embedding_style} on the denoiser")
    else:
        if tc.norm_order_style not in _hip.ORDER or tc.final_norm_order_style not in (None,) + tuple(_hip.ORDER):
            bad.append(f"norm order {tc.norm_order_style} / {tc.final_norm_order_style}")
        if tc.pos_embedding_style not in _POS_KIND:
            bad.append(f"pos_embedding_style={tc.pos_embedding_style}")
    if not tc.use_swiglu and tc.ffn_inner_activation_name not in _hip.ACT:      # unknown to the reference too
        bad.append(f"ffn activation {tc.ffn_inner_activation_name}")
    if bad:
        raise NotImplementedError(f"{name}: not implemented on the HIP path: {', '.join(bad)}")


def final_norm_absent(tc: TransformerConfig) -> bool:
    """A contextualizer / base-LCM stack ends without a LayerNorm when its final order is "post"
    (lcm/models/two_tower_diffusion_lcm/builder.py:583-595, lcm/nn/transformer/decoder.py:119-122)."""
    return (tc.final_norm_order_style or tc.norm_order_style) == "post"


def extra_layer_spec(spec: list, p: str, tc: TransformerConfig, d: int, norm: Callable[[str], None]) -> None:
    """Per-layer tensors of the non-default layer options of a contextualizer / base-LCM stack: the normformer order's
    norm after the attention block (the stack's norm style) and fairseq2's LayerNorm (weight + bias) on the two-matrix
    FFN's inner activations; the learned residual scale of the FFN block, initialised at `scale_residual`
    (lcm/nn/transformer/factory.py:172-184)."""
    if tc.norm_order_style == "normformer":
        norm(f"{p}.self_attn_norm")
        if not tc.use_swiglu:
            F = tc.allocated_ffn_dim()
            spec.append((f"{p}.ffn.inner_layer_norm.weight", (F,), "ones"))
            spec.append((f"{p}.ffn.inner_layer_norm.bias", (F,), "zeros"))
    if tc.scale_residual is not None:
        spec.append((f"{p}.residual_scale", (d,), f"const:{float(tc.scale_residual)}"))


def sinusoidal_table(n: int, dim: int, *, device, dtype) -> Tensor:
    """fairseq2 SinusoidalPositionEncoder's table (as built at lcm/nn/transformer/factory.py:199-204): fp32 rows
    [sin(p f_0) .. sin(p f_{h-1}), cos(p f_0) .. cos(p f_{h-1})], h = dim / 2, f_i = exp(-i ln(10000) / (h - 1)).  The
    contextualizer adds these rows on the fly in fp32 (LCMHIP_POS_SINE); for the denoiser's latents (position 0 in the
    sampling loop, position j in the all-positions forward) the table is handed to the library in the operand dtype, in
    the place of learned positions -- a buffer of the module, not a state-dict entry, as in the reference."""
    h = dim // 2
    steps = torch.arange(n, dtype=torch.float32, device=device)
    freqs = torch.exp(torch.arange(h, dtype=torch.float32, device=device) * (-math.log(10000.0) / (h - 1)))
    ang = torch.outer(steps, freqs)
    return torch.cat([ang.sin(), ang.cos()], dim=-1).to(dtype).contiguous()


def prelu_spec(spec: list, p: str, tc: TransformerConfig) -> None:
    """torch.nn.PReLU() as the two-matrix FFN's inner activation module (lcm/nn/initialization.py:135-136): one learned
    slope, initialised at 0.25."""
    if not tc.use_swiglu and tc.ffn_inner_activation_name == "prelu":
        spec.append((f"{p}.ffn.inner_activation.weight", (1,), "const:0.25"))


def tower_desc(tc: TransformerConfig, denoiser: bool = False) -> "_hip.TowerDesc":
    """lcmhip_tower_desc of one stack."""
    return _hip.TowerDesc(
        num_layers=tc.num_layers, num_heads=tc.num_attn_heads, ffn_dim=tc.allocated_ffn_dim(),
        norm_kind=_NORM_KIND[tc.layer_normalization_style],
        ffn_kind=_hip.FFN_SWIGLU if tc.use_swiglu else _hip.FFN_RELU,
        # the denoiser's sinusoidal positions reach the library as a table (see sinusoidal_table)
        pos_kind=_hip.POS_LEARNED if denoiser and tc.pos_embedding_style == "sine" else _POS_KIND[tc.pos_embedding_style],
        out_proj_bias=int(tc.mha_output_proj_bias),
        rope_theta=float(tc.rope_theta), qk_layernorm=int(tc.enable_qk_layernorm),
        ffn_activation=_hip.ACT.get(tc.ffn_inner_activation_name, 0),
        norm_order=0 if denoiser else _hip.ORDER[tc.norm_order_style],
        no_final_norm=0 if denoiser else int(final_norm_absent(tc)),
        scale_residual=0 if denoiser else int(tc.scale_residual is not None))


def normalizer_config_for(name: Optional[str], dim: int) -> SonarNormalizerConfig:
    """Config of the normalizer a model names (reference: abstract_lcm/builder.py:89-101 loads that card, so the
    model's normalizer has the card's arch, clipping included).  Unknown or absent names give the base arch."""
    from lcm.models.sonar_normalizer.builder import sonar_normalizer_archs
    from lcm.utils.card_utils import resolve_card
    cfg = SonarNormalizerConfig(dim=dim)
    if name is not None:
        try:
            card = resolve_card(name)
        except ValueError:
            card = None
        if card is not None:
            arch = sonar_normalizer_archs[card.get("model_arch", "base")]
            cfg = SonarNormalizerConfig(**{**arch.__dict__, "dim": dim})
    return cfg


def load_named_normalizer_statistics(normalizer: SonarNormalizer, name: Optional[str]) -> bool:
    """The reference builds a model around the normalizer its config names, loaded from that normalizer's card with its
    fitted statistics (abstract_lcm/builder.py:89-101), so a freshly built model already normalizes with them; a
    checkpoint's own `sonar_normalizer.*` tensors replace them afterwards.  Here: if `name` is a card with a checkpoint of
    matching shapes, its tensors are copied into the model's normalizer; otherwise the identity statistics stay."""
    if name is None:
        return False
    from lcm.utils.card_utils import resolve_card, resolve_checkpoint_path
    try:
        path = resolve_checkpoint_path(resolve_card(name))
    except ValueError:
        return False
    if path is None or not os.path.exists(path):
        return False
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob["model"] if "model" in blob else blob
    own = normalizer.state_dict()
    if any(k not in sd or tuple(sd[k].shape) != tuple(t.shape) for k, t in own.items()):
        return False
    normalizer.load_state_dict(sd)
    return True


def normalizer_spec(prefix: str, ncfg: SonarNormalizerConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    D = ncfg.dim
    spec = [(f"{prefix}.center", (D,), "zeros"), (f"{prefix}.scale", (D,), "ones")]
    if ncfg.clip_proba is not None:      # sonar_normalizer/builder.py:111-117: both start at one
        spec += [(f"{prefix}.clip_min", (D,), "ones"), (f"{prefix}.clip_max", (D,), "ones")]
    return spec


def parameter_spec(config: TwoTowerDiffusionLCModelConfig,
                   normalizer: Optional[SonarNormalizerConfig] = None) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) of every parameter/buffer, named like the reference's state dict."""
    d, D = config.model_dim, config.sonar_embed_dim
    spec: List[Tuple[str, Tuple[int, ...], str]] = normalizer_spec("sonar_normalizer", normalizer or SonarNormalizerConfig(dim=D)) + [
        ("encoder_frontend.pre_linear.weight", (d, D), "matrix")]
    if config.frontend.pre_linear_bias:                 # frontend.py:24-36, 96-104
        spec.append(("encoder_frontend.pre_linear.bias", (d,), "zeros"))
    if config.context_encoder.pos_embedding_style == "learned":
        spec.append(("encoder_frontend.pos_encoder.weight", (config.max_seq_len, d), "pos"))

    def stack(prefix: str, tc: TransformerConfig, denoiser: bool) -> None:
        F = tc.allocated_ffn_dim()
        ln_weight = tc.layer_normalization_style != "unit"       # lcm/nn/normalization.py:56-70: no learnable parameters
        ln_bias = tc.layer_normalization_style == "standard"      # "rms" and "fp32" are built with bias=False

        def norm(n: str) -> None:
            if ln_weight:
                spec.append((f"{n}.weight", (d,), "ones"))
            if ln_bias:
                spec.append((f"{n}.bias", (d,), "zeros"))

        for i in range(tc.num_layers):
            p = f"{prefix}.layers.{i}"
            norm(f"{p}.self_attn_layer_norm")
            if denoiser:
                norm(f"{p}.cross_attention_layer_norm")
            norm(f"{p}.ffn_layer_norm")
            for attn in (["self_attn", "cross_attention"] if denoiser else ["self_attn"]):
                for proj in ("q_proj", "k_proj", "v_proj"):
                    spec.append((f"{p}.{attn}.{proj}.weight", (d, d), "matrix"))
                    spec.append((f"{p}.{attn}.{proj}.bias", (d,), "zeros"))
                spec.append((f"{p}.{attn}.output_proj.weight", (d, d), "matrix"))
                if tc.mha_output_proj_bias:
                    spec.append((f"{p}.{attn}.output_proj.bias", (d,), "zeros"))
                if tc.enable_qk_layernorm:       # lcm/nn/transformer/attention.py:131-141
                    norm(f"{p}.{attn}.q_layer_norm")
                    norm(f"{p}.{attn}.k_layer_norm")
            if tc.use_swiglu:
                spec.append((f"{p}.ffn.gate_proj.weight", (F, d), "matrix"))
                spec.append((f"{p}.ffn.gate_proj.bias", (F,), "zeros"))
            spec.append((f"{p}.ffn.inner_proj.weight", (F, d), "matrix"))
            spec.append((f"{p}.ffn.inner_proj.bias", (F,), "zeros"))
            spec.append((f"{p}.ffn.output_proj.weight", (d, F), "matrix"))
            spec.append((f"{p}.ffn.output_proj.bias", (d,), "zeros"))
            prelu_spec(spec, p, tc)
            if denoiser:
                spec.append((f"{p}.modulator.fc.weight", (9 * d, d), "modulator"))
                spec.append((f"{p}.modulator.fc.bias", (9 * d,), "modulator"))
            else:
                extra_layer_spec(spec, p, tc, d, norm)
        if denoiser or not final_norm_absent(tc):
            norm(f"{prefix}.layer_norm")

    stack("context_encoder", config.context_encoder, denoiser=False)
    spec += [
        ("denoiser.embed_time.fc1.weight", (d, TIMESTEP_FREQ_DIM), "matrix"), ("denoiser.embed_time.fc1.bias", (d,), "zeros"),
        ("denoiser.embed_time.fc2.weight", (d, d), "matrix"), ("denoiser.embed_time.fc2.bias", (d,), "zeros"),
        ("denoiser.initial_proj.fc.weight", (d, D), "matrix"), ("denoiser.final_proj.fc.weight", (D, d), "matrix"),
    ]
    if config.denoiser.pre_denoiser.linear_bias:         # lcm/nn/projection.py:28, 60-67
        spec.append(("denoiser.initial_proj.fc.bias", (d,), "zeros"))
    if config.denoiser.post_denoiser.linear_bias:
        spec.append(("denoiser.final_proj.fc.bias", (D,), "zeros"))
    if config.denoiser.pos_embedding_style == "learned":       # lcm/nn/transformer/factory.py:187-198
        spec.append(("denoiser.pos_encoder.weight", (config.max_seq_len, d), "pos"))
    stack("denoiser", config.denoiser, denoiser=True)
    return spec


class DeviceSession:
    """Caches and scratch of one batch of documents inside a torch-owned workspace."""

    def __init__(self, model: "TwoTowerDiffusionLCModel", max_batch: int, max_ctx: int, max_steps: int,
                 keep_context: bool = False) -> None:
        self.model = model
        self.dims = (max_batch, max_ctx, max_steps)
        # options of the session (lcmhip.h LCMHIP_SESSION_*): the model's, plus the fp32 copy of the contextualizer
        # output that encode() without a state bag reads back
        self.flags = int(getattr(model, "session_flags", 0)) | (_hip.SESSION_KEEP_CONTEXT if keep_context else 0)
        nbytes = C.c_size_t(0)
        _hip.check(_hip.lib.lcmhip_session_workspace_bytes(model._handle, max_batch, max_ctx, max_steps, self.flags, C.byref(nbytes)))
        self.workspace = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=model.device)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        self._h = C.c_void_p()
        self._parked = None
        # lcmhip_session_create writes its tables into the workspace before it returns, outside stream order: the
        # allocator may have handed out a block that work still pending on this stream uses until it has run
        torch.cuda.current_stream(model.device).synchronize()
        _hip.check(_hip.lib.lcmhip_session_create(model._handle, max_batch, max_ctx, max_steps, self.flags, C.c_void_p(base),
                                                  nbytes.value, C.byref(self._h)))
        self.schedule_key = None

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _hip is not None and getattr(_hip, "lib", None) is not None:   # None during interpreter shutdown
            _hip.lib.lcmhip_session_destroy(h)

    def _stream(self):
        return _hip.stream_ptr(self.model.device)

    def park(self) -> None:
        """Called when the session goes back to the model's pool: marks the point in the releasing stream behind
        which its caches are idle."""
        self._parked = torch.cuda.Event()
        self._parked.record(torch.cuda.current_stream(self.model.device))

    def unpark(self) -> None:
        """Called when a pooled session is taken again, possibly by another host thread on another stream: that
        stream waits for the work the session was released with, and the allocator learns that the workspace is
        in use there."""
        cur = torch.cuda.current_stream(self.model.device)
        if self._parked is not None:
            cur.wait_event(self._parked)
            self._parked = None
        self.workspace.record_stream(cur)

    def set_schedule(self, timesteps: Optional[List[int]] = None) -> None:
        """Sampling schedule of the model's scheduler, or (with `timesteps`) only the adaLN table of the given
        diffusion timesteps for `denoise_positions` (no guidance, sampling scalars unused)."""
        m = self.model
        sch = m.noise_scheduler
        if timesteps is None:
            key = m._schedule_key()
            ts = [int(t) for t in sch.timesteps.tolist()]
            params = _hip.DenoiseParams(
                guidance_scale=float(m.guidance_scale), guidance_rescale=float(m.guidance_rescale),
                ddim_eta=float(m.ddim_eta), epsilon_scaling=float(m.epsilon_scaling) if m.epsilon_scaling else 0.0,
                clip_noise=float(m.clip_noise), init_noise_sigma=float(sch.init_noise_sigma),
                prediction_type=_hip.PRED[sch.prediction_type], clip_sample=int(sch.config.clip_sample),
                clip_sample_range=float(sch.config.clip_sample_range), thresholding=int(sch.config.thresholding),
                dynamic_thresholding_ratio=float(sch.config.dynamic_thresholding_ratio),
                sample_max_value=float(sch.config.sample_max_value))
        else:
            ts = [int(t) for t in timesteps]
            key = ("table", tuple(ts))
            params = _hip.DenoiseParams(
                guidance_scale=1.0, guidance_rescale=0.0, ddim_eta=0.0, epsilon_scaling=0.0, clip_noise=100.0,
                init_noise_sigma=1.0, prediction_type=_hip.PRED[sch.prediction_type], clip_sample=0, clip_sample_range=1.0,
                thresholding=0, dynamic_thresholding_ratio=0.995, sample_max_value=1.0)
        if key == self.schedule_key:
            return
        step_ratio = sch.step_ratio if timesteps is None else 1
        ts_arr = (C.c_int32 * len(ts))(*ts)
        ac = sch.alphas_cumprod.tolist()
        ac_arr = (C.c_float * len(ac))(*ac)
        self.schedule_key = None      # a rejected call leaves the session without a usable schedule of this key
        _hip.check(_hip.lib.lcmhip_session_set_schedule(
            self._h, ts_arr, len(ts), ac_arr, len(ac), float(sch.final_alpha_cumprod), step_ratio,
            C.byref(params), self._stream()))
        self.schedule_key = key

    def reset(self, n_docs: int) -> None:
        _hip.check(_hip.lib.lcmhip_session_reset(self._h, n_docs, self._stream()))

    # A with_fft normalizer (sonar_normalizer/builder.py:121-137) transforms raw embeddings before center / scale / clip
    # and transforms back after them.  The library's fused normalize / denormalize steps do the center / scale / clip
    # part; the transform itself (lcmhip_rfft_pack) runs here, on the way in and on the way out.
    def _with_fft(self) -> bool:
        n = getattr(self.model, "sonar_normalizer", None)
        return n is not None and n.config.with_fft

    def _to_model_space(self, x32: Tensor, normalized: bool) -> Tensor:
        return _hip.rfft_pack(x32, inverse=False) if (not normalized and self._with_fft()) else x32

    def _from_model_space(self, out: Optional[Tensor]) -> Optional[Tensor]:
        return _hip.rfft_pack(out, inverse=True) if (out is not None and self._with_fft()) else out

    def append(self, x: Tensor, normalized: bool) -> None:
        """x: (n_active, s, D) embeddings of the next s context positions."""
        _hip.require_gpu(x, "context embeddings")
        x32 = self._to_model_space(x.float().contiguous(), normalized)
        assert x32.dim() == 3 and x32.shape[0] == self.num_active, (tuple(x32.shape), self.num_active)
        _hip.check(_hip.lib.lcmhip_context_append(self._h, _hip.ptr(x32), x32.shape[1], int(normalized), self._stream()))

    def base_forward(self, x: Tensor, *, normalized: bool, noise: Optional[Tensor], temperature: float,
                     all_positions: bool, want_output: bool = True) -> Optional[Tensor]:
        """Base LCM: appends the s positions of x (n_active, s, D) and returns the predictions, fp32, for every
        position (n_active, s, D) or for the last one (n_active, D); None when only the cache is wanted.  Raw embeddings
        in, raw predictions out; with `normalized` both sides stay in the normalizer's space (the model's `forward`)."""
        _hip.require_gpu(x, "input embeddings")
        x32 = self._to_model_space(x.float().contiguous(), normalized)
        assert x32.dim() == 3 and x32.shape[0] == self.num_active, (tuple(x32.shape), self.num_active)
        out = None
        if want_output:
            shape = x32.shape if all_positions else (x32.shape[0], x32.shape[2])
            out = torch.empty(shape, dtype=torch.float32, device=x32.device)
        nz = None
        if noise is not None and want_output:
            nz = noise.float().contiguous()
            assert nz.numel() == out.numel(), "noise must have the shape of the output"
        _hip.check(_hip.lib.lcmhip_base_forward(self._h, _hip.ptr(x32), x32.shape[1], int(normalized), _hip.ptr(nz),
                                                float(temperature), int(all_positions), _hip.ptr(out), self._stream()))
        return out if normalized else self._from_model_space(out)      # normalized in, normalized out

    def denoise(self, noise: Tensor, trace: Optional[Tensor] = None, variance_noise: Optional[Tensor] = None) -> Tensor:
        """noise: (n_active, D) N(0, 1) draws -> next sentence embedding (n_active, D) fp32.  `variance_noise`
        (n_steps, n_active, D): the per-step draws of stochastic DDIM (needed when the schedule has ddim_eta > 0)."""
        _hip.require_gpu(noise, "noise")
        self.set_schedule()      # a no-op unless a `denoise_positions` table (model.denoise / forward) replaced the sampling schedule
        nz = noise.float().contiguous()
        assert nz.shape[0] == self.num_active
        out = torch.empty_like(nz)
        if variance_noise is None:
            _hip.check(_hip.lib.lcmhip_denoise(self._h, _hip.ptr(nz), _hip.ptr(out), _hip.ptr(trace), self._stream()))
            return self._from_model_space(out)
        vn = variance_noise.float().contiguous()
        assert vn.dim() == 3 and vn.shape[1:] == nz.shape, (tuple(vn.shape), tuple(nz.shape))
        _hip.check(_hip.lib.lcmhip_denoise_stochastic(self._h, _hip.ptr(nz), _hip.ptr(vn), _hip.ptr(out), _hip.ptr(trace),
                                                      self._stream()))
        return self._from_model_space(out)

    def denoise_positions(self, noisy: Tensor, step_index: Tensor, source_lengths: Optional[Tensor] = None,
                          context_lengths: Optional[Tensor] = None) -> Tensor:
        """noisy (n_active, T, D) latents in normalized space, step_index (n_active, T) indices into the table set by
        `set_schedule(timesteps)`; optional per-document source / context lengths -> model output (n_active, T, D) fp32."""
        _hip.require_gpu(noisy, "noisy latents")
        x32 = noisy.float().contiguous()
        assert x32.dim() == 3 and x32.shape[0] == self.num_active, (tuple(x32.shape), self.num_active)
        idx = step_index.to(device=x32.device, dtype=torch.int32).contiguous()
        assert tuple(idx.shape) == tuple(x32.shape[:2]), (tuple(idx.shape), tuple(x32.shape))

        def per_doc(t: Optional[Tensor]) -> Optional[Tensor]:
            if t is None:
                return None
            t = t.to(device=x32.device, dtype=torch.int32).contiguous()
            assert tuple(t.shape) == (x32.shape[0],), tuple(t.shape)
            return t

        src, ctx = per_doc(source_lengths), per_doc(context_lengths)
        out = torch.empty_like(x32)
        _hip.check(_hip.lib.lcmhip_denoise_positions(self._h, _hip.ptr(x32), _hip.ptr(idx), x32.shape[1], _hip.ptr(src),
                                                     _hip.ptr(ctx), _hip.ptr(out), self._stream()))
        return out

    def reorder(self, order) -> None:
        idx = [int(i) for i in (order.tolist() if isinstance(order, Tensor) else order)]
        if idx == list(range(self.num_active)):
            return
        arr = (C.c_int32 * len(idx))(*idx)
        _hip.check(_hip.lib.lcmhip_session_reorder(self._h, arr, len(idx), self._stream()))

    @property
    def context_len(self) -> int:
        return _hip.lib.lcmhip_session_context_len(self._h)

    @property
    def num_active(self) -> int:
        return _hip.lib.lcmhip_session_num_active(self._h)

    def read_context(self, row: int) -> Tensor:
        out = torch.empty(self.context_len, self.model.model_dim, dtype=torch.float32, device=self.model.device)
        _hip.check(_hip.lib.lcmhip_session_read_context(self._h, row, _hip.ptr(out), self._stream()))
        return out

    def profile_enable(self, on: bool) -> None:
        _hip.check(_hip.lib.lcmhip_session_profile_enable(self._h, int(on)))

    def profile(self) -> List[dict]:
        rows = (_hip.ProfileRow * 64)()
        n = C.c_int32(0)
        _hip.check(_hip.lib.lcmhip_session_profile(self._h, rows, 64, C.byref(n), self._stream()))
        return [dict(name=rows[i].name.decode(), launches=rows[i].launches, total_ms=rows[i].total_ms,
                     bytes=rows[i].bytes, flops=rows[i].flops) for i in range(n.value)]


class DeviceContext:
    """Handle of the contextualizer output kept on the device (what the reference passes around as
    the `context` EmbeddingsBatch).  `seqs` reads it back when the session keeps a copy
    (`session_flags` with SESSION_KEEP_CONTEXT)."""

    def __init__(self, session: Optional[DeviceSession], length: int) -> None:
        self.session = session
        self.length = length
        self.padding_mask = None

    @property
    def seqs(self) -> Tensor:
        s = self.session
        return torch.stack([s.read_context(i) for i in range(s.num_active)]).to(s.model.dtype)


class TwoTowerDiffusionLCModel(AbstractLCModel):
    config: TwoTowerDiffusionLCModelConfig

    def __init__(self, config: TwoTowerDiffusionLCModelConfig, *, device=None, dtype: Optional[torch.dtype] = None,
                 seed: Optional[int] = None) -> None:
        super().__init__(config)
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("TwoTowerDiffusionLCModel runs only on a GPU (HIP path); there is no CPU fallback")
        dtype = dtype or torch.float32
        _hip.dtype_code(dtype)
        _check_supported("context_encoder", config.context_encoder)
        _check_supported("denoiser", config.denoiser, denoiser=True)
        if config.denoiser.timestep_embed_dim != config.model_dim:
            raise ValueError("timestep_embed_dim must equal model_dim")
        for which in ("pre_denoiser", "post_denoiser"):
            if getattr(config.denoiser, which).activation_name is not None:
                # projection.py:73-86; the sampling loop folds these projections' split-K slabs into its neighbours
                raise NotImplementedError(f"denoiser.{which}.activation_name is not implemented on the HIP path")
        self.model_dim = config.model_dim
        self.sonar_embed_dim = config.sonar_embed_dim
        ncfg = normalizer_config_for(config.sonar_normalizer_name, config.sonar_embed_dim)
        self._spec = parameter_spec(config, ncfg)
        self._params: Dict[str, Tensor] = {
            name: torch.empty(shape, dtype=dtype, device=device) for name, shape, _ in self._spec}
        # derived tensors: the denoiser's self-attention always sees one position, where softmax == 1 and the
        # block collapses to output_proj(v_proj(h)); W_ov = W_o W_v and b_ov = W_o b_v + b_o are folded once per
        # weight load (by the library's own GEMM) so that a diffusion step streams one matrix instead of two
        d = config.model_dim
        self._derived: Dict[str, Tensor] = {}
        for i in range(config.denoiser.num_layers):
            self._derived[f"denoiser.layers.{i}.self_attn.ov_proj.weight"] = torch.empty((d, d), dtype=dtype, device=device)
            self._derived[f"denoiser.layers.{i}.self_attn.ov_proj.bias"] = torch.empty((d,), dtype=dtype, device=device)
        # a projection configured without a bias (frontend.pre_linear_bias, pre_denoiser / post_denoiser.linear_bias) has
        # none in the state dict; the library takes a zero vector in its place
        for name, n in (("encoder_frontend.pre_linear.bias", d), ("denoiser.initial_proj.fc.bias", d),
                        ("denoiser.final_proj.fc.bias", config.sonar_embed_dim)):
            if name not in self._params:
                self._derived[name] = torch.zeros((n,), dtype=dtype, device=device)
        if config.denoiser.pos_embedding_style == "sine":
            self._derived["denoiser.pos_encoder.weight"] = sinusoidal_table(config.max_seq_len, d, device=device, dtype=dtype)
        self._handle = None
        self.init_parameters(seed if seed is not None else 0)
        self.sonar_normalizer = SonarNormalizer(ncfg, device=device, dtype=dtype)
        # the normalizer's buffers ARE the model's tensors (one storage, like a registered submodule)
        self.sonar_normalizer.center = self._params["sonar_normalizer.center"]
        self.sonar_normalizer.scale = self._params["sonar_normalizer.scale"]
        if ncfg.clip_proba is not None:
            self.sonar_normalizer.clip_min = self._params["sonar_normalizer.clip_min"]
            self.sonar_normalizer.clip_max = self._params["sonar_normalizer.clip_max"]
        load_named_normalizer_statistics(self.sonar_normalizer, config.sonar_normalizer_name)
        self.noise_scheduler = DDIMScheduler(config.noise_scheduler)
        self._handle = C.c_void_p()
        self._bind()
        self._refresh_derived()
        self._session_pool: List[DeviceSession] = []
        self._session_lock = threading.Lock()      # generators on several host threads share one model
        self.session_flags = 0     # _hip.SESSION_* bits given to every session this model creates (tests, profiling)

    # ---- parameters ---------------------------------------------------------------------------------
    def init_parameters(self, seed: int = 0, matrix_std: float = 0.02, modulator_std: float = 0.0) -> None:
        """Seeded initialisation: N(0, matrix_std) matrices, zero biases, unit norms, N(0, 1)
        learned positions (frontend.py:113-125) and zero adaLN modulators (lcm_denoiser.py:55-58)
        unless `modulator_std` > 0."""
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
                std = {"matrix": matrix_std, "pos": self.config.frontend.embedding_std, "modulator": modulator_std}[kind]
                if std == 0.0:
                    t.zero_()
                else:
                    t.copy_(torch.randn(shape, generator=g, device=t.device, dtype=torch.float32) * std)
        self._refresh_derived()

    def _fold_versions(self) -> Tuple[int, ...]:
        """In-place edit counters of the tensors the folded self-attention weights are computed from."""
        out = []
        for i in range(self.config.denoiser.num_layers):
            p = f"denoiser.layers.{i}.self_attn"
            for k in (f"{p}.output_proj.weight", f"{p}.output_proj.bias", f"{p}.v_proj.weight", f"{p}.v_proj.bias"):
                t = self._params.get(k)
                out.append(-1 if t is None else t._version)
        return tuple(out)

    def _ensure_folded(self) -> None:
        """state_dict() hands out the live parameter tensors; if one that W_ov / b_ov derive from was edited in place
        since the last fold (copy_, a perturbation in a test, ...), fold again before the weights are used."""
        if getattr(self, "_folded_versions", None) != self._fold_versions():
            self._refresh_derived()

    def _refresh_derived(self) -> None:
        if not getattr(self, "_handle", None):
            return
        P, code, st = self._params, _hip.dtype_code(self.dtype), _hip.stream_ptr(self.device)
        d = self.config.model_dim
        for i in range(self.config.denoiser.num_layers):
            p = f"denoiser.layers.{i}.self_attn"
            wo, wv_t = P[f"{p}.output_proj.weight"], P[f"{p}.v_proj.weight"].t().contiguous()
            bo = P.get(f"{p}.output_proj.bias")
            # W_ov[i][j] = sum_k W_o[i][k] W_v[k][j]  ==  lcmhip_gemm(A = W_o, W = W_v^T)
            _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(wo), d, _hip.ptr(wv_t), None, None, None, None,
                                            _hip.ptr(self._derived[f"{p}.ov_proj.weight"]), d, d, d, d, _hip.EPI_STORE,
                                            code, None, 0, st))
            # b_ov = W_o b_v + b_o
            _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(P[f"{p}.v_proj.bias"]), d, _hip.ptr(wo), None, _hip.ptr(bo), None, None,
                                            _hip.ptr(self._derived[f"{p}.ov_proj.bias"]), d, 1, d, d, _hip.EPI_STORE,
                                            code, None, 0, st))
        torch.cuda.current_stream(self.device).synchronize()   # wv_t temporaries may be freed after this
        self._folded_versions = self._fold_versions()

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
        self._refresh_derived()

    def num_parameters(self) -> int:
        return sum(t.numel() for n, t in self._params.items() if not n.startswith("sonar_normalizer."))

    def model_desc(self) -> "_hip.ModelDesc":
        """The lcmhip_model_desc this model was created from."""
        c = self.config
        return _hip.ModelDesc(sonar_dim=c.sonar_embed_dim, model_dim=c.model_dim, max_seq_len=c.max_seq_len,
                              timestep_freq_dim=TIMESTEP_FREQ_DIM, dtype=_hip.dtype_code(self.dtype), norm_eps=1e-5,
                              context=tower_desc(c.context_encoder), denoiser=tower_desc(c.denoiser, denoiser=True))

    def bound_tensors(self) -> Dict[str, Tensor]:
        """Every tensor bound into the library's weight table, by name: the state dict plus the folded self-attention
        products derived from it."""
        self._ensure_folded()
        return {**self._params, **self._derived}

    def _bind(self) -> None:
        desc = self.model_desc()
        _hip.check(_hip.lib.lcmhip_model_create(C.byref(desc), C.byref(self._handle)))
        code = _hip.dtype_code(self.dtype)
        for name, t in list(self._params.items()) + list(self._derived.items()):
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

    # ---- generation set-up ------------------------------------------------------------------------------
    def prep_for_denoising(self, decoding_options) -> None:
        o = decoding_options
        self.guidance_scale = o.guidance_scale
        self.guidance_rescale = o.guidance_rescale
        self.initial_noise_scale = o.initial_noise_scale
        self.timesteps = o.inference_timesteps
        self.clip_noise = o.clip_noise
        self.ddim_eta = o.ddim_eta
        self.epsilon_scaling = o.epsilon_scaling
        self.do_classifier_free_guidance = self.guidance_scale != 1.0
        self.noise_scheduler.set_timesteps(self.timesteps, device=self.device)
        self.noise_scheduler.init_noise_sigma = self.initial_noise_scale
        if o.thresholding:          # reference builder.py:234-242: the options override the scheduler's config
            self.noise_scheduler.config.thresholding = o.thresholding
            self.noise_scheduler.config.dynamic_thresholding_ratio = o.dynamic_thresholding_ratio
            self.noise_scheduler.config.sample_max_value = o.sample_max_value

    def _schedule_key(self):
        s = self.noise_scheduler
        c = s.config
        return (tuple(s.timesteps.tolist()), self.guidance_scale, self.guidance_rescale, self.ddim_eta,
                self.epsilon_scaling, self.clip_noise, s.init_noise_sigma, s.prediction_type,
                c.clip_sample, c.clip_sample_range, c.thresholding, c.dynamic_thresholding_ratio, c.sample_max_value)

    def acquire_session(self, max_batch: int, max_ctx: int, timesteps: Optional[List[int]] = None) -> DeviceSession:
        if timesteps is None:
            assert hasattr(self, "clip_noise"), (
                "The model is not properly set for decoding, make sure to call `model.prep_for_denoising()`")
            steps = len(self.noise_scheduler.timesteps)
        else:
            steps = len(timesteps)
        with self._session_lock:
            self._ensure_folded()
            sess = None
            for i, s in enumerate(self._session_pool):
                if s.dims[0] >= max_batch and s.dims[1] >= max_ctx and s.dims[2] >= steps and s.flags == int(self.session_flags):
                    sess = self._session_pool.pop(i)
                    break
            else:
                self._session_pool.clear()   # one workspace at a time: they are large
            s = None                         # the loop variable would keep the dropped session's workspace alive below
        if sess is None:
            sess = DeviceSession(self, max_batch, max_ctx, steps)
        else:
            sess.unpark()
        sess.set_schedule(timesteps)
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

    # ---- reference API ------------------------------------------------------------------------------------
    def sample_initial_noise_vectors(self, batch_size: int) -> Tensor:
        """N(0, 1) draws for the next sentence; the scheduler's sigma and the clip are applied by
        the denoising kernel (reference: builder.py:244-260 returns them already scaled)."""
        assert hasattr(self, "clip_noise"), (
            "The model is not properly set for decoding, make sure to call `model.prep_for_denoising()`")
        return torch.randn(batch_size, 1, self.config.sonar_embed_dim, device=self.device)

    @torch.inference_mode()
    def forward(self, batch: EmbeddingsBatch, noisy_batch: EmbeddingsBatch, cf_guidance_prob: float = 0.0) -> EmbeddingsBatch:
        """The reference's training-shaped forward (builder.py:128-159) in inference mode: contextualize the clean
        `batch` (normalized embeddings, optional padding mask, optional `source_lengths`), then denoise every position
        of `noisy_batch` (latents + per-position `diffusion_timesteps`) against the causally shifted context.
        Returns the model output (B, T, D) fp32 with the noisy batch's padding mask."""
        if cf_guidance_prob != 0.0:
            raise NotImplementedError("cf_guidance_prob > 0 drops random conditioning rows (training only); not implemented on the HIP path")
        x, noisy, ts = batch.seqs, noisy_batch.seqs, noisy_batch.diffusion_timesteps
        assert ts is not None, "noisy_batch.diffusion_timesteps is required"
        assert x.shape[0] == noisy.shape[0]
        ts = ts.reshape(noisy.shape[0], noisy.shape[1])
        uniq, inverse = torch.unique(ts, sorted=True, return_inverse=True)
        sess = self.acquire_session(x.shape[0], x.shape[1], timesteps=uniq.tolist())
        try:
            sess.reset(x.shape[0])
            sess.append(x, normalized=True)
            ctx_lens = batch.padding_mask.seq_lens if batch.padding_mask is not None else None
            out = sess.denoise_positions(noisy, inverse, batch.source_lengths, ctx_lens)
        finally:
            self.release_session(sess)
        return EmbeddingsBatch(out, noisy_batch.padding_mask)

    @torch.inference_mode()
    def denoise(self, noisy_batch: EmbeddingsBatch, context: DeviceContext, source_lengths: Optional[Tensor] = None,
                cf_guidance_prob: float = 0.0, state_bag: Optional[LCMIncrementalStateBag] = None,
                inference: bool = False) -> EmbeddingsBatch:
        """One pass of the denoiser (reference builder.py:193-213; lcm_denoiser.py:152-251): `noisy_batch.seqs`
        (B, T, D) latents in normalized space with their `diffusion_timesteps` (B, T), conditioned on `context`, the
        handle `encode` returned for the same B documents.  With `inference` every latent sees the zero vector and the
        whole context (attention_masks.py:139-140: no cross-attention mask); without it position j sees context
        positions [0, j + source_length) (the shifted causal mask of the training-shaped forward).  Self-attention is
        diagonal either way.  Returns the model output (B, T, D) fp32; what it means is the scheduler's
        `prediction_type`.  `state_bag` is accepted for the reference's signature; the denoiser keeps no state."""
        if cf_guidance_prob != 0.0:
            raise NotImplementedError("cf_guidance_prob > 0 drops random conditioning rows (training only); not implemented on the HIP path")
        sess = getattr(context, "session", None)
        assert sess is not None, "denoise expects the context handle returned by `encode`"
        noisy, ts = noisy_batch.seqs, noisy_batch.diffusion_timesteps
        assert ts is not None, "noisy_batch.diffusion_timesteps is required"
        assert noisy.dim() == 3 and noisy.shape[0] == sess.num_active, (tuple(noisy.shape), sess.num_active)
        ts = ts.reshape(noisy.shape[0], noisy.shape[1])
        uniq, inverse = torch.unique(ts, sorted=True, return_inverse=True)
        sess.set_schedule(uniq.tolist())      # adaLN table of these timesteps; the sampling loop sets its own again
        if inference:
            source_lengths = torch.full((noisy.shape[0],), sess.context_len, dtype=torch.int32, device=noisy.device)
        out = sess.denoise_positions(noisy, inverse, source_lengths, None)
        return EmbeddingsBatch(out, noisy_batch.padding_mask)

    @torch.inference_mode()
    def apply_classifier_free_guidance(self, predicted_noise: Tensor) -> Tensor:
        """Classifier-free guidance with rescale (reference builder.py:463-488): `predicted_noise` holds the
        conditional rows in its first half and the unconditional rows in its second; returns the guided half.  This is
        the arithmetic the sampling loop applies inside its fused DDIM step, reached here through that kernel's
        entry point: an epsilon-type step from x_t = 0 to alpha_prev = 0 returns the guided prediction unchanged."""
        assert hasattr(self, "guidance_scale"), (
            "The model is not properly set for decoding, make sure to call `model.prep_for_denoising()`")
        _hip.require_gpu(predicted_noise, "predicted noise")
        cond, uncond = predicted_noise.float().chunk(2)
        D = cond.shape[-1]
        c2, u2 = cond.reshape(-1, D).contiguous(), uncond.reshape(-1, D).contiguous()
        guided = torch.empty_like(c2)
        params = _hip.DenoiseParams(
            guidance_scale=float(self.guidance_scale), guidance_rescale=float(self.guidance_rescale), ddim_eta=0.0,
            epsilon_scaling=0.0, clip_noise=float("inf"), init_noise_sigma=1.0, prediction_type=_hip.PRED["epsilon"],
            clip_sample=0, clip_sample_range=1.0, thresholding=0, dynamic_thresholding_ratio=0.995, sample_max_value=1.0)
        _hip.check(_hip.lib.lcmhip_ddim_cfg_step(_hip.ptr(torch.zeros_like(c2)), _hip.ptr(c2), _hip.ptr(u2), _hip.ptr(guided),
                                                 None, c2.shape[0], D, 0.5, 0.0, C.byref(params), None,
                                                 _hip.stream_ptr(c2.device)))
        return guided.reshape(cond.shape).to(predicted_noise.dtype)

    @torch.inference_mode()
    def encode(self, batch: EmbeddingsBatch, state_bag: Optional[LCMIncrementalStateBag] = None,
               *, normalized: bool = True, **kwargs) -> DeviceContext:
        """Contextualizes `batch.seqs` (normalized SONAR embeddings) after what the bag already holds."""
        if batch.padding_mask is not None:
            raise ValueError("the contextualizer takes unpadded positions (the generator never pads them)")
        if state_bag is None:
            # reference builder.py:161-191 without a bag: a plain forward over the whole sequence.  A throw-away
            # session that keeps the contextualizer output, so that `.seqs` of the result can be read back.
            B, S = batch.seqs.shape[:2]
            # room for the adaLN table of a `denoise` call on this context: as many timesteps as the sampling schedule
            steps = len(self.noise_scheduler.timesteps) if getattr(self.noise_scheduler, "timesteps", None) is not None else 1
            sess = DeviceSession(self, B, S, max(1, steps), keep_context=True)
            sess.reset(B)
            sess.append(batch.seqs, normalized=normalized)
            return DeviceContext(sess, sess.context_len)
        sess = self._session_for(state_bag, batch.seqs.shape[0])
        sess.append(batch.seqs, normalized=normalized)
        return DeviceContext(sess, sess.context_len)

    @torch.inference_mode()
    def predict_next_sentence(self, batch: EmbeddingsBatch, context: DeviceContext, temperature: float = 1.0,
                              state_bag: Optional[LCMIncrementalStateBag] = None,
                              context_state_bag: Optional[LCMIncrementalStateBag] = None,
                              noise: Optional[Tensor] = None, trace: Optional[Tensor] = None,
                              variance_noise: Optional[Tensor] = None, **kwargs) -> Tuple[EmbeddingsBatch, DeviceContext]:
        """batch.seqs (B, 1, D): the latest raw SONAR embedding of every active document.
        Returns the next embedding (B, 1, D) fp32 and the grown context handle.  `noise` (B, 1, D) or
        (B, D) replaces the internal N(0, 1) draw; `variance_noise` (n_steps, B, D) the per-step draws of
        stochastic DDIM (ddim_eta > 0)."""
        assert context_state_bag is not None, "Expected a state_bag to incrementally encode the context"
        x = batch.seqs
        sess = self._session_for(context_state_bag, x.shape[0])
        sess.append(x, normalized=False)           # normalizer fused into the staging kernel
        context_state_bag.increment_step_nr(x.shape[1])
        if noise is None:
            noise = self.sample_initial_noise_vectors(x.shape[0])
        if self.ddim_eta > 0 and variance_noise is None:
            # the reference draws these inside DDIMScheduler.step (randn_tensor, ddim.py:495-503), one per step
            variance_noise = torch.randn(len(self.noise_scheduler.timesteps), x.shape[0], self.config.sonar_embed_dim,
                                         device=self.device)
        if variance_noise is not None:
            variance_noise = variance_noise.reshape(len(self.noise_scheduler.timesteps), x.shape[0], -1)
        out = sess.denoise(noise.reshape(x.shape[0], -1), trace, variance_noise if self.ddim_eta > 0 else None)
        return EmbeddingsBatch(out.unsqueeze(1), None), DeviceContext(sess, sess.context_len)

    # classifier-free guidance is selected by `prep_for_denoising`; both entry points share the kernel loop
    predict_next_sentence_with_cf_guidance = predict_next_sentence


class TwoTowerDiffusionLCModelBuilder(AbstractLCModelBuilder):
    """The reference's builder class (builder.py:491-608) as far as it applies here: `build_model` (and the inherited
    `build_sonar_normalizer`).  Its `build_frontend / build_context_encoder / build_denoiser / build_noise_scheduler` make
    torch modules, which this implementation does not have (the model is a table of device tensors bound to the C
    library)."""

    config: TwoTowerDiffusionLCModelConfig

    def build_model(self) -> "TwoTowerDiffusionLCModel":
        return TwoTowerDiffusionLCModel(self.config, device=self.device, dtype=self.dtype)

    def build_noise_scheduler(self) -> DDIMScheduler:
        return DDIMScheduler(self.config.noise_scheduler)


def create_two_tower_diffusion_lcm_model(config: TwoTowerDiffusionLCModelConfig, *, device=None,
                                         dtype: Optional[torch.dtype] = None,
                                         seed: Optional[int] = None) -> TwoTowerDiffusionLCModel:
    return TwoTowerDiffusionLCModel(config, device=device, dtype=dtype, seed=seed)
