"""Seeded synthetic weights with the reference's state-dict names (test infrastructure).

Names follow the module attribute names of lcm/models/two_tower_diffusion_lcm/builder.py:104-121,
lcm/models/two_tower_diffusion_lcm/frontend.py:80-104, lcm/nn/denoisers/lcm_denoiser.py:123-141,
310-379, lcm/nn/timestep_encoder.py:43-57, lcm/nn/projection.py:60-71 and fairseq2's
StandardMultiheadAttention / FeedForwardNetwork / StandardTransformerDecoderLayer attribute names
(q_proj, k_proj, v_proj, output_proj; gate_proj, inner_proj, output_proj;
self_attn_layer_norm, ffn_layer_norm).

Trained checkpoints are not available offline, so every matrix -- *including the adaLN
modulators*, which the reference zero-initialises (lcm_denoiser.py:55-58) and which would make
every denoiser layer the identity -- is drawn from a seeded normal distribution.
"""

import hashlib
import os
from typing import Dict

import torch

from .archs import ArchConfig, TowerConfig

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_dummy_normalizer(path: str = None) -> Dict[str, torch.Tensor]:
    """center/scale of the reference's only binary fixture
    (lcm/cards/mock_data/dummy_normalizer.pt, `{"model": {"center","scale"}}`), kept as data under
    tests/golden/."""
    path = path or os.path.join(_GOLDEN, "dummy_normalizer.pt")
    blob = torch.load(path, weights_only=True)
    m = blob["model"]
    return {"center": m["center"].float().clone(), "scale": m["scale"].float().clone()}


def _seed_for(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little") % (2**63 - 1)


def _normal(seed: int, name: str, shape, std: float) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_for(seed, name))
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def _tower_shapes(prefix: str, d: int, tc: TowerConfig, cross: bool, modulator: bool):
    """Yields (name, shape, kind) for one stack of layers; kind in {w, b, norm_w, norm_b, mod_w, mod_b}."""
    F = tc.ffn_dim()
    for i in range(tc.num_layers):
        p = f"{prefix}.layers.{i}"
        norms = ["self_attn_layer_norm"] + (["cross_attention_layer_norm"] if cross else []) + ["ffn_layer_norm"]
        for n in norms:
            if tc.norm != "unit":            # lcm/nn/normalization.py:56-70: no learnable parameters
                yield f"{p}.{n}.weight", (d,), "norm_w"
            if tc.norm == "standard":
                yield f"{p}.{n}.bias", (d,), "norm_b"
        for attn in ["self_attn"] + (["cross_attention"] if cross else []):
            for proj in ("q_proj", "k_proj", "v_proj"):
                kind = "qk_w" if proj in ("q_proj", "k_proj") else "w"
                yield f"{p}.{attn}.{proj}.weight", (d, d), kind
                yield f"{p}.{attn}.{proj}.bias", (d,), "b"
            yield f"{p}.{attn}.output_proj.weight", (d, d), "w"
            if tc.out_proj_bias:
                yield f"{p}.{attn}.output_proj.bias", (d,), "b"
            if tc.qk_layernorm:              # attention.py:131-141: the tower's own norm style over model_dim
                for n in ("q_layer_norm", "k_layer_norm"):
                    if tc.norm != "unit":
                        yield f"{p}.{attn}.{n}.weight", (d,), "norm_w"
                    if tc.norm == "standard":
                        yield f"{p}.{attn}.{n}.bias", (d,), "norm_b"
        if tc.use_swiglu:
            yield f"{p}.ffn.gate_proj.weight", (F, d), "w"
            yield f"{p}.ffn.gate_proj.bias", (F,), "b"
        yield f"{p}.ffn.inner_proj.weight", (F, d), "w"
        yield f"{p}.ffn.inner_proj.bias", (F,), "b"
        yield f"{p}.ffn.output_proj.weight", (d, F), "w"
        yield f"{p}.ffn.output_proj.bias", (d,), "b"
        if not tc.use_swiglu and tc.activation == "prelu":
            yield f"{p}.ffn.inner_activation.weight", (1,), "prelu"
        if modulator:
            yield f"{p}.modulator.fc.weight", (9 * d, d), "mod_w"
            yield f"{p}.modulator.fc.bias", (9 * d,), "mod_b"
        elif not cross:
            # contextualizer / base-LCM stacks: the normformer order's extra norms and the learned residual scale
            # (lcm/nn/transformer/factory.py:165-184, 292-300)
            if getattr(tc, "norm_order", "pre") == "normformer":
                if tc.norm != "unit":
                    yield f"{p}.self_attn_norm.weight", (d,), "norm_w"
                if tc.norm == "standard":
                    yield f"{p}.self_attn_norm.bias", (d,), "norm_b"
                if not tc.use_swiglu:
                    yield f"{p}.ffn.inner_layer_norm.weight", (F,), "norm_w"
                    yield f"{p}.ffn.inner_layer_norm.bias", (F,), "norm_b"
            if getattr(tc, "scale_residual", False):
                yield f"{p}.residual_scale", (d,), "res_scale"
    final_post = (getattr(tc, "final_norm_order", None) or getattr(tc, "norm_order", "pre")) == "post"
    if not cross and final_post:
        return
    if tc.norm != "unit":
        yield f"{prefix}.layer_norm.weight", (d,), "norm_w"
    if tc.norm == "standard":
        yield f"{prefix}.layer_norm.bias", (d,), "norm_b"


def state_dict_spec(cfg: ArchConfig):
    """(name, shape, kind) for every tensor of the model, in a fixed order."""
    d, D = cfg.model_dim, cfg.sonar_dim
    yield "sonar_normalizer.center", (D,), "normalizer"
    yield "sonar_normalizer.scale", (D,), "normalizer"
    yield "encoder_frontend.pre_linear.weight", (d, D), "w"
    if getattr(cfg, "frontend_bias", True):
        yield "encoder_frontend.pre_linear.bias", (d,), "b"
    if cfg.context.pos == "learned":
        yield "encoder_frontend.pos_encoder.weight", (cfg.max_seq_len, d), "pos"
    yield from _tower_shapes("context_encoder", d, cfg.context, cross=False, modulator=False)
    yield "denoiser.embed_time.fc1.weight", (d, cfg.timestep_freq_dim), "t_w"
    yield "denoiser.embed_time.fc1.bias", (d,), "b"
    yield "denoiser.embed_time.fc2.weight", (d, d), "t_w"
    yield "denoiser.embed_time.fc2.bias", (d,), "b"
    yield "denoiser.initial_proj.fc.weight", (d, D), "w"
    if getattr(cfg, "initial_proj_bias", True):
        yield "denoiser.initial_proj.fc.bias", (d,), "b"
    yield "denoiser.final_proj.fc.weight", (D, d), "w"
    if getattr(cfg, "final_proj_bias", True):
        yield "denoiser.final_proj.fc.bias", (D,), "b"
    if cfg.denoiser.pos == "learned":         # lcm/nn/transformer/factory.py:187-198 through denoisers/factory.py:143
        yield "denoiser.pos_encoder.weight", (cfg.max_seq_len, d), "pos"
    yield from _tower_shapes("denoiser", d, cfg.denoiser, cross=True, modulator=True)


def base_state_dict_spec(cfg):
    """(name, shape, kind) of a base LCM (module names of lcm/models/base_lcm/builder.py:84-88,
    base_lcm/frontend.py:96-124, lcm/nn/projection.py:66-69)."""
    d, D = cfg.model_dim, cfg.sonar_dim
    if cfg.sonar_normalizer_name is not None:
        yield "frontend.sonar_normalizer.center", (D,), "normalizer"
        yield "frontend.sonar_normalizer.scale", (D,), "normalizer"
    yield "frontend.pre_linear.weight", (d, D), "w"
    if cfg.pre_linear_bias:
        yield "frontend.pre_linear.bias", (d,), "b"
    if cfg.lcm.pos == "learned":
        yield "frontend.pos_encoder.weight", (cfg.max_seq_len, d), "pos"
    yield from _tower_shapes("lcm", d, cfg.lcm, cross=False, modulator=False)
    yield "postnet.fc.weight", (D, d), "w"
    if cfg.postnet_bias:
        yield "postnet.fc.bias", (D,), "b"


def num_parameters(cfg: ArchConfig) -> int:
    n = 0
    for name, shape, kind in state_dict_spec(cfg):
        if kind == "normalizer":
            continue
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


def init_state_dict(cfg: ArchConfig, seed: int = 0, *, w_std: float = 0.02, b_std: float = 0.02,
                    qk_gain: float = 1.0, mod_gain: float = 1.0, t_gain: float = 1.0,
                    pos_std: float = 1.0, norm_jitter: float = 0.0,
                    normalizer: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state dict.

    `qk_gain` scales q/k projection weights (attention logits grow with its square) so that
    softmax / masking errors cannot hide behind near-uniform attention; `mod_gain` scales the adaLN
    modulator weights so gates/scales/shifts are O(0.1-1) instead of O(0.01); `t_gain` scales the
    timestep-MLP weights; `norm_jitter` perturbs norm weights away from 1.
    Each tensor has its own generator seeded from (seed, name): values do not depend on order.
    """
    normalizer = normalizer or load_dummy_normalizer()
    sd: Dict[str, torch.Tensor] = {}
    spec = base_state_dict_spec(cfg) if hasattr(cfg, "lcm") else state_dict_spec(cfg)
    for name, shape, kind in spec:
        if kind == "normalizer":
            sd[name] = normalizer[name.split(".")[-1]].clone()
        elif kind == "w":
            sd[name] = _normal(seed, name, shape, w_std)
        elif kind == "qk_w":
            sd[name] = _normal(seed, name, shape, w_std * qk_gain)
        elif kind == "t_w":
            sd[name] = _normal(seed, name, shape, w_std * t_gain)
        elif kind == "mod_w":
            sd[name] = _normal(seed, name, shape, w_std * mod_gain)
        elif kind == "mod_b":
            sd[name] = _normal(seed, name, shape, b_std * mod_gain)
        elif kind == "b":
            sd[name] = _normal(seed, name, shape, b_std)
        elif kind == "pos":
            sd[name] = _normal(seed, name, shape, pos_std)
        elif kind == "norm_w":
            sd[name] = 1.0 + _normal(seed, name, shape, norm_jitter)
        elif kind == "norm_b":
            sd[name] = _normal(seed, name, shape, norm_jitter)
        elif kind == "prelu":                 # away from PReLU's initial 0.25 and from LeakyReLU's 0.01
            sd[name] = 0.4 + _normal(seed, name, shape, 0.05)
        elif kind == "res_scale":             # around 0.5, away from the neutral 1
            sd[name] = 0.5 + _normal(seed, name, shape, 0.1)
        else:
            raise AssertionError(kind)
    return sd


def synthetic_prompts(normalizer: Dict[str, torch.Tensor], batch: int, length: int, seed: int = 1234) -> torch.Tensor:
    """Prompts x = center + scale * z, z ~ N(0, 1): SONAR-like vectors whose normalised form is unit-normal."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(batch, length, normalizer["center"].numel(), generator=g, dtype=torch.float32)
    return normalizer["center"] + normalizer["scale"] * z


def synthetic_noise(steps: int, batch: int, dim: int = 1024, seed: int = 4321) -> torch.Tensor:
    """Host-supplied initial noise, consumed as noise[generation_step, document]."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(steps, batch, dim, generator=g, dtype=torch.float32)
