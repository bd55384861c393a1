"""Two-tower diffusion LCM restated in plain PyTorch-CPU (test infrastructure).

Follows lcm/models/two_tower_diffusion_lcm/builder.py:161-488 (encode / denoise /
prep_for_denoising / predict_next_sentence[_with_cf_guidance] / apply_classifier_free_guidance),
lcm/models/two_tower_diffusion_lcm/frontend.py:127-152, lcm/nn/transformer/decoder.py:29-176,
lcm/nn/transformer/attention.py:143-298, lcm/nn/denoisers/lcm_denoiser.py:33-64, 152-251,
388-546, lcm/nn/timestep_encoder.py:72-122, lcm/nn/normalization.py:16-88,
lcm/models/sonar_normalizer/builder.py:121-137, plus the fairseq2 0.3 semantics those files
rely on (RotaryEncoder, LearnedPositionEncoder, RMSNorm, GLU/Standard FFN, TorchSDPA,
CausalAttentionMaskFactory, FullAttentionState), which are restated from their published
behaviour because fairseq2 is not installable here.
"""

import dataclasses
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .archs import ArchConfig, TowerConfig
from .scheduler import DDIMSchedule


# ----------------------------------------------------------------------------- primitives

# lcm/nn/initialization.py:121-151 (parse_activation_fn): the torch.nn modules of these names with default arguments,
# the inner activation of the two-matrix FFN (transformer/factory.py:289-300).  "prelu" (a learned slope) is left out.
ACTIVATIONS = {
    "relu": torch.nn.ReLU(), "tanh": torch.nn.Tanh(), "elu": torch.nn.ELU(), "leaky_relu": torch.nn.LeakyReLU(),
    "selu": torch.nn.SELU(), "gelu": torch.nn.GELU(), "silu": torch.nn.SiLU(), "softsign": torch.nn.Softsign(),
    "sigmoid": torch.nn.Sigmoid(), "hardsigmoid": torch.nn.Hardsigmoid(),
}

def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, kind: str, eps: float) -> Tensor:
    """lcm/nn/normalization.py:73-88.  rms = fairseq2 RMSNorm(bias=False): normalise in fp32,
    cast back, then multiply by weight; standard = F.layer_norm with weight and bias; fp32 = layer_norm
    evaluated in fp32 with a weight and no bias (:17-33, 46-53); unit = layer_norm without parameters (:56-70)."""
    if kind == "unit":
        return F.layer_norm(x, (x.shape[-1],), None, None, eps)
    w = sd[prefix + ".weight"]
    if kind == "fp32":
        return F.layer_norm(x.float(), (x.shape[-1],), w.float(), None, eps).type_as(x)
    if kind == "rms":
        xf = x.float()
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        return y.type_as(x) * w
    if kind == "standard":
        return F.layer_norm(x, (x.shape[-1],), w, sd[prefix + ".bias"], eps)
    raise ValueError(kind)


def rope(x: Tensor, start: int, theta: float) -> Tensor:
    """fairseq2 RotaryEncoder: consecutive pairs (x[2i], x[2i+1]) rotated by pos * theta^(-2i/hd),
    computed in fp32 through complex multiplication.  x: (N, H, S, hd)."""
    hd = x.shape[-1]
    S = x.shape[-2]
    idx = torch.arange(0, hd, 2, dtype=torch.float32)
    freqs = 1.0 / (theta ** (idx / hd))
    pos = torch.arange(start, start + S, dtype=torch.float32)
    ang = torch.outer(pos, freqs)
    rot = torch.polar(torch.ones_like(ang), ang)                       # (S, hd/2) complex
    xc = torch.view_as_complex(x.float().unflatten(-1, (-1, 2)).contiguous())
    return torch.view_as_real(xc * rot).flatten(-2).type_as(x)


def rope_explicit(x: Tensor, start: int, theta: float) -> Tensor:
    """Second, independent statement of the same rotation (cos/sin on interleaved pairs), used by
    the tests to cross-check `rope`."""
    hd = x.shape[-1]
    S = x.shape[-2]
    out = torch.empty_like(x, dtype=torch.float32)
    xf = x.float()
    for s in range(S):
        for i in range(hd // 2):
            a = (start + s) * theta ** (-2.0 * i / hd)
            c, sn = math.cos(a), math.sin(a)
            x0, x1 = xf[..., s, 2 * i], xf[..., s, 2 * i + 1]
            out[..., s, 2 * i] = x0 * c - x1 * sn
            out[..., s, 2 * i + 1] = x0 * sn + x1 * c
    return out.type_as(x)


def sdpa(q: Tensor, k: Tensor, v: Tensor, key_lens: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """fairseq2 TorchSDPA semantics with explicit softmax: scale 1/sqrt(hd); padded keys and
    masked positions get -inf.  q (N,H,S,hd), k/v (N,H,Sk,hd); key_lens (N,) number of visible
    keys per row (None = all); mask (S,Sk) additive float or None."""
    hd = q.shape[-1]
    w = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (hd**-0.5)
    if mask is not None:
        w = w + mask
    if key_lens is not None:
        Sk = k.shape[-2]
        pad = torch.arange(Sk)[None, :] >= key_lens[:, None]                # True = padded
        w = w.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(w, dim=-1)
    return torch.matmul(p, v.float()).type_as(q)


def fft_transform(x: Tensor) -> Tensor:
    """FFTInterface.fft_transform, lcm/models/sonar_normalizer/builder.py:59-67."""
    X = torch.fft.rfft(x.float(), norm="backward")
    return torch.cat([X.real, X.imag[..., 1:-1]], dim=-1).to(x.dtype)


def fft_inverse_transform(y: Tensor) -> Tensor:
    """FFTInterface.fft_inverse_transform, lcm/models/sonar_normalizer/builder.py:69-84."""
    n = y.shape[-1]
    rr, im = torch.split(y.float(), [n // 2 + 1, n // 2 - 1], dim=-1)
    z = torch.zeros_like(im[..., :1])
    return torch.fft.irfft(torch.complex(rr, torch.cat([z, im, z], dim=-1))).to(y.dtype)


def sinusoidal_positions(start: int, n: int, dim: int) -> Tensor:
    """fairseq2 SinusoidalPositionEncoder as lcm/nn/transformer/factory.py:199-204 builds it (no legacy padding
    offset): an fp32 table, row p = [sin(p f_0) .. sin(p f_{h-1}), cos(p f_0) .. cos(p f_{h-1})] with h = dim / 2 and
    f_i = exp(-i ln(10000) / (h - 1)); the encoder adds it to the sequence in fp32 and casts back."""
    h = dim // 2
    steps = torch.arange(start, start + n, dtype=torch.float32)
    freqs = torch.exp(torch.arange(h, dtype=torch.float32) * (-math.log(10000.0) / (h - 1)))
    ang = torch.outer(steps, freqs)
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def causal_mask(attn_len: int, key_len: int) -> Optional[Tensor]:
    """fairseq2 CausalAttentionMaskFactory in eval: None when a single position is fed
    (incremental decode attends to every cached key); otherwise the last `attn_len` rows of a
    key_len x key_len lower-triangular mask."""
    if attn_len <= 1:
        return None
    full = torch.full((key_len, key_len), float("-inf")).triu(1)
    return full[key_len - attn_len:, :]


class KVCache:
    """fairseq2 IncrementalStateBag + FullAttentionState for the contextualizer
    (lcm/nn/incremental_state.py:14-43; lcm/nn/transformer/attention.py:227-253)."""

    def __init__(self, max_num_steps: int):
        self.max_num_steps = max_num_steps
        self.step_nr = 0
        self.k: Dict[int, Tensor] = {}
        self.v: Dict[int, Tensor] = {}

    def increment_step_nr(self, n: int = 1) -> None:
        if self.step_nr + n > self.max_num_steps:
            raise ValueError(f"step_nr {self.step_nr}+{n} exceeds max_num_steps {self.max_num_steps}")
        self.step_nr += n

    def append(self, layer: int, k: Tensor, v: Tensor) -> Tuple[Tensor, Tensor]:
        if layer in self.k:
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        else:
            self.k[layer], self.v[layer] = k, v
        return self.k[layer], self.v[layer]

    def reorder(self, new_order: Tensor) -> None:
        for d in (self.k, self.v):
            for key in d:
                d[key] = d[key].index_select(0, new_order)


# ----------------------------------------------------------------------------- model

class OracleTwoTower:
    def __init__(self, cfg: ArchConfig, sd: Dict[str, Tensor]):
        self.cfg = cfg
        self.sd = sd
        self.model_dim = cfg.model_dim
        self.scheduler = DDIMSchedule(dataclasses.replace(cfg.scheduler))   # own copy: prep_for_denoising edits it
        # Tests with long contexts set this: the cross-attention k/v projections of the context are then computed
        # once per predict_next_sentence call instead of at every diffusion step (the context does not change
        # between the steps of one call, so the values are the same; the reference, and the CPU baseline timing,
        # recompute them every step).
        self.memoize_cross_kv = False
        self._kv_memo = None

    # -- SONAR normalizer (lcm/models/sonar_normalizer/builder.py:121-137; clipping when the state dict holds
    #    clip_min / clip_max, i.e. the "clipping" arch; no fft)
    def normalize(self, x: Tensor) -> Tensor:
        if getattr(self.cfg, "normalizer_fft", False):       # builder.py:121-123
            x = fft_transform(x)
        y = (x - self.sd["sonar_normalizer.center"]) / self.sd["sonar_normalizer.scale"]
        if "sonar_normalizer.clip_min" in self.sd:
            y = torch.clamp(y, min=self.sd["sonar_normalizer.clip_min"], max=self.sd["sonar_normalizer.clip_max"])
        return y

    def denormalize(self, x: Tensor) -> Tensor:
        if "sonar_normalizer.clip_min" in self.sd:
            x = torch.clamp(x, min=self.sd["sonar_normalizer.clip_min"], max=self.sd["sonar_normalizer.clip_max"])
        y = x * self.sd["sonar_normalizer.scale"] + self.sd["sonar_normalizer.center"]
        return fft_inverse_transform(y) if getattr(self.cfg, "normalizer_fft", False) else y   # builder.py:134-137

    # -- attention / ffn blocks
    def _mha(self, prefix: str, tc: TowerConfig, x_q: Tensor, x_kv: Tensor, *, key_lens=None, mask=None,
             cache: Optional[KVCache] = None, layer: int = -1, use_rope: bool = False) -> Tensor:
        """lcm/nn/transformer/attention.py:201-298."""
        sd, H = self.sd, tc.num_heads
        qk_ln = getattr(tc, "qk_layernorm", False)          # attention.py:151-155, 178-182: before the head split
        q = linear(x_q, sd, prefix + ".q_proj")
        if qk_ln:
            q = norm(q, sd, prefix + ".q_layer_norm", tc.norm, self.cfg.norm_eps)
        q = q.unflatten(-1, (H, -1)).transpose(1, 2)
        memo = getattr(self, "_kv_memo", None) if x_kv is not x_q else None
        if memo is not None and prefix in memo:
            k, v = memo[prefix]
        else:
            k = linear(x_kv, sd, prefix + ".k_proj")
            if qk_ln:
                k = norm(k, sd, prefix + ".k_layer_norm", tc.norm, self.cfg.norm_eps)
            k = k.unflatten(-1, (H, -1)).transpose(1, 2)
            v = linear(x_kv, sd, prefix + ".v_proj").unflatten(-1, (H, -1)).transpose(1, 2)
            if memo is not None:
                memo[prefix] = (k, v)
        if use_rope:
            start = cache.step_nr if cache is not None else 0
            q = rope(q, start, tc.rope_theta)
            k = rope(k, start, tc.rope_theta)
        if cache is not None:
            k, v = cache.append(layer, k, v)
        o = sdpa(q, k, v, key_lens, mask)
        o = o.transpose(1, 2).flatten(2, 3)
        return linear(o, sd, prefix + ".output_proj")

    def _ffn(self, prefix: str, tc: TowerConfig, x: Tensor) -> Tensor:
        sd = self.sd
        if tc.use_swiglu:   # fairseq2 GLUFeedForwardNetwork: output(inner(x) * silu(gate(x)))
            return linear(linear(x, sd, prefix + ".inner_proj") * F.silu(linear(x, sd, prefix + ".gate_proj")),
                          sd, prefix + ".output_proj")
        h = linear(x, sd, prefix + ".inner_proj")
        if tc.activation == "prelu":      # torch.nn.PReLU(): one learned slope, a parameter of the activation module
            h = F.prelu(h, sd[prefix + ".inner_activation.weight"])
        else:
            h = ACTIVATIONS[tc.activation](h)
        if prefix + ".inner_layer_norm.weight" in sd:
            # fairseq2 StandardFeedForwardNetwork built with the normformer order (factory.py:292-300): its default
            # LayerNorm (weight + bias) on the inner activations, before the output projection
            h = norm(h, sd, prefix + ".inner_layer_norm", "standard", self.cfg.norm_eps)
        return linear(h, sd, prefix + ".output_proj")

    # -- contextualizer (builder.py:161-191)
    def encode(self, x_norm: Tensor, cache: Optional[KVCache] = None, *, frontend: str = "encoder_frontend",
               decoder: str = "context_encoder", tc: Optional[TowerConfig] = None) -> Tensor:
        cfg, sd = self.cfg, self.sd
        tc = tc or self.cfg.context
        S = x_norm.shape[1]
        start = cache.step_nr if cache is not None else 0
        x = linear(x_norm, sd, frontend + ".pre_linear")                     # frontend.py:139
        if tc.pos == "learned":                                             # frontend.py:141-147
            x = x + sd[frontend + ".pos_encoder.weight"][start:start + S]
        elif tc.pos == "sine":
            x = (x.float() + sinusoidal_positions(start, S, x.shape[-1])).to(x.dtype)
        mask = causal_mask(S, start + S)                                     # decoder.py:145-152
        order = getattr(tc, "norm_order", "pre")
        final_order = getattr(tc, "final_norm_order", None) or order
        for i in range(tc.num_layers):
            p = f"{decoder}.layers.{i}"
            # attention block: decoder.py:55-88 (no residual scale here)
            h = x if order == "post" else norm(x, sd, p + ".self_attn_layer_norm", tc.norm, cfg.norm_eps)
            a = self._mha(p + ".self_attn", tc, h, h, mask=mask, cache=cache, layer=i, use_rope=(tc.pos == "rope"))
            if order == "normformer":
                a = norm(a, sd, p + ".self_attn_norm", tc.norm, cfg.norm_eps)
            x = a + x
            if order == "post":
                x = norm(x, sd, p + ".self_attn_layer_norm", tc.norm, cfg.norm_eps)
            # FFN block: fairseq2 StandardTransformerDecoderLayer._forward_ffn, which LCMStandardTransformerDecoderLayer
            # inherits -- the one place the learned residual scale is applied
            h = x if order == "post" else norm(x, sd, p + ".ffn_layer_norm", tc.norm, cfg.norm_eps)
            f = self._ffn(p + ".ffn", tc, h)
            if getattr(tc, "scale_residual", False):
                x = sd[p + ".residual_scale"] * x
            x = f + x
            if order == "post":
                x = norm(x, sd, p + ".ffn_layer_norm", tc.norm, cfg.norm_eps)
        if final_order == "post":                                            # decoder.py:119-122, 170-171
            return x
        return norm(x, sd, decoder + ".layer_norm", tc.norm, cfg.norm_eps)

    # -- timestep embedding (lcm/nn/timestep_encoder.py:72-122)
    def embed_time(self, timesteps: Tensor) -> Tensor:
        half = self.cfg.timestep_freq_dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        args = timesteps.reshape(-1, 1).float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = F.silu(linear(emb, self.sd, "denoiser.embed_time.fc1"))
        h = linear(h, self.sd, "denoiser.embed_time.fc2")
        return h.view(*timesteps.shape, self.cfg.model_dim)

    # -- denoiser (lcm_denoiser.py:152-251, 388-546)
    def denoise(self, latents: Tensor, timesteps: Tensor, context: Tensor,
                ctx_lens: Optional[Tensor] = None, cross_mask: Optional[Tensor] = None) -> Tensor:
        """latents (N,S,D), timesteps (N,S) long, context (N,t,d); ctx_lens (N,) visible context
        length per row or None.  Diagonal self mask, key padding from ctx_lens shifted by the prepended
        zero vector; cross_mask None = inference path (S = 1), else the additive (N,1,S,t+1) or (S,t+1)
        shifted causal mask of the training-shaped forward."""
        cfg, sd, tc = self.cfg, self.sd, self.cfg.denoiser
        emb_t = self.embed_time(timesteps)                                           # :184
        cond = torch.cat([torch.zeros_like(context[:, 0:1]), context], dim=1)        # :193-199
        key_lens = None if ctx_lens is None else ctx_lens + 1                        # :201-206
        x = linear(latents, sd, "denoiser.initial_proj.fc")                          # :209
        S = x.shape[1]
        if tc.pos == "learned":                                                      # :211-212, no state bag: from 0
            x = x + sd["denoiser.pos_encoder.weight"][:S]
        elif tc.pos == "sine":                                                       # the same call with the sinusoidal encoder
            x = (x.float() + sinusoidal_positions(0, S, x.shape[-1])).to(x.dtype)
        self_mask = torch.eye(S).log()                                               # attention_masks.py:110-113
        for i in range(tc.num_layers):
            p = f"denoiser.layers.{i}"
            mod = linear(F.silu(emb_t), sd, p + ".modulator.fc")                     # :60-64
            m_san, m_x, m_ffn = mod.chunk(3, dim=-1)
            # self attention block (:431-473)
            shift, scale, gate = m_san.chunk(3, dim=-1)
            h = norm(x, sd, p + ".self_attn_layer_norm", tc.norm, cfg.norm_eps) * (1 + scale) + shift
            x = x + gate * self._mha(p + ".self_attn", tc, h, h, mask=self_mask, use_rope=(tc.pos == "rope"))
            # cross attention block (:475-521)
            shift, scale, gate = m_x.chunk(3, dim=-1)
            h = norm(x, sd, p + ".cross_attention_layer_norm", tc.norm, cfg.norm_eps) * (1 + scale) + shift
            x = x + gate * self._mha(p + ".cross_attention", tc, h, cond, key_lens=key_lens, mask=cross_mask,
                                     use_rope=False)
            # ffn block (:523-546)
            shift, scale, gate = m_ffn.chunk(3, dim=-1)
            h = norm(x, sd, p + ".ffn_layer_norm", tc.norm, cfg.norm_eps) * (1 + scale) + shift
            x = x + gate * self._ffn(p + ".ffn", tc, h)
        x = norm(x, sd, "denoiser.layer_norm", tc.norm, cfg.norm_eps)                # :243-244
        return linear(x, sd, "denoiser.final_proj.fc")                               # :249

    def forward(self, x_norm: Tensor, noisy: Tensor, timesteps: Tensor, lengths: Optional[Tensor] = None,
                source_lengths: Optional[Tensor] = None) -> Tensor:
        """TwoTowerDiffusionLCModel.forward, builder.py:128-159, with cf_guidance_prob = 0: contextualize the clean
        normalized batch x_norm (N,T,D), then denoise noisy (N,S,D) with per-position timesteps (N,S) under the
        ShiftedCausalAttentionMask (lcm/nn/denoisers/attention_masks.py:21-46, 196-228): query i sees key j of
        [zero vector, context] iff j <= i + source_length.  `lengths` (N,) = valid context positions (padding mask)."""
        context = self.encode(x_norm)
        S, K = noisy.shape[1], context.shape[1] + 1
        if source_lengths is None:
            mask = torch.ones(S, K).tril_(diagonal=0).log()
        else:
            mask = torch.stack([torch.ones(S, K).tril_(diagonal=int(n)) for n in source_lengths]).unsqueeze(1).log()
        return self.denoise(noisy, timesteps, context, ctx_lens=lengths, cross_mask=mask)

    # -- generation-time set-up (builder.py:215-242)
    def prep_for_denoising(self, opt) -> None:
        self.guidance_scale = opt.guidance_scale
        self.guidance_rescale = opt.guidance_rescale
        self.initial_noise_scale = opt.initial_noise_scale
        self.clip_noise = opt.clip_noise
        self.ddim_eta = opt.ddim_eta
        self.epsilon_scaling = opt.epsilon_scaling
        self.do_cfg = self.guidance_scale != 1.0
        self.scheduler.set_timesteps(opt.inference_timesteps)
        self.scheduler.init_noise_sigma = self.initial_noise_scale
        if opt.thresholding:
            self.scheduler.cfg.thresholding = True
            self.scheduler.cfg.dynamic_thresholding_ratio = opt.dynamic_thresholding_ratio
            self.scheduler.cfg.sample_max_value = opt.sample_max_value

    def initial_latents(self, noise: Tensor) -> Tensor:
        """builder.py:244-260 with the N(0,1) draw supplied by the caller: scale by sigma, clip."""
        lat = noise.float() * self.scheduler.init_noise_sigma
        return lat.clip(-self.clip_noise, self.clip_noise)

    def apply_cfg(self, eps: Tensor) -> Tensor:
        # builder.py:463-488 (torch.std default: unbiased)
        c, u = eps.chunk(2)
        g = u + self.guidance_scale * (c - u)
        std_pos = c.std(dim=-1, keepdim=True)
        std_cfg = g.std(dim=-1, keepdim=True)
        factor = self.guidance_rescale * (std_pos / std_cfg) + (1 - self.guidance_rescale)
        return factor * g

    def predict_next_sentence(self, x: Tensor, context: Tensor, cache: KVCache, noise: Tensor,
                              trace: Optional[List] = None, variance_noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """x (B,1,D) raw SONAR embeddings, context (B,t,d) -> (next embedding (B,1,D), new context).
        builder.py:262-338 (no guidance) and :340-461 (classifier-free guidance).  `noise` (B,1,D)
        replaces the in-function torch.randn; `variance_noise` (steps,B,1,D) the per-step draws of DDIMScheduler.step
        when eta > 0 (ddim.py:495-503); `trace`, if given, collects per-step latents."""
        def vnoise(si):
            return None if variance_noise is None else variance_noise[si].reshape(noise.shape)
        sch = self.scheduler
        new_ctx = self.encode(self.normalize(x), cache)
        cache.increment_step_nr(1)
        context = torch.cat((context, new_ctx), dim=1)
        latents = self.initial_latents(noise)
        B = x.shape[0]
        self._kv_memo = {} if self.memoize_cross_kv else None      # valid for this call's (fixed) context only
        if not self.do_cfg:
            for si, t in enumerate(sch.timesteps.tolist()):
                ts = torch.full((B, 1), t, dtype=torch.long)
                out = self.denoise(latents, ts, context)
                prev, x0 = sch.step(out, t, latents, eta=self.ddim_eta, epsilon_scaling=self.epsilon_scaling,
                                    variance_noise=vnoise(si))
                latents = prev.clip(-self.clip_noise, self.clip_noise)
                if trace is not None:
                    trace.append((t, latents.clone(), x0.clone()))
            self._kv_memo = None
            return self.denormalize(x0), context

        t_len = context.shape[1]
        ctx2 = torch.cat((context, torch.zeros_like(context)), dim=0)                 # :378-381
        lens2 = torch.cat((torch.full((B,), t_len), torch.zeros(B, dtype=torch.long)))
        for si, t in enumerate(sch.timesteps.tolist()):
            is_max = t == sch.num_train_steps - 1                                     # :385-387
            inp = torch.cat([latents, latents], dim=0)
            ts = torch.full((2 * B, 1), t, dtype=torch.long)
            out = self.denoise(inp, ts, ctx2, lens2)
            if is_max:                                                                # :404-419
                prev, x0 = sch.step(out.chunk(2)[0], t, latents, eta=self.ddim_eta,
                                    epsilon_scaling=self.epsilon_scaling, variance_noise=vnoise(si))
            else:                                                                     # :420-449
                eps = sch.get_epsilon(out, inp, t)
                eps = self.apply_cfg(eps)
                prev, x0 = sch.step(eps, t, latents, eta=self.ddim_eta,
                                    epsilon_scaling=self.epsilon_scaling, prediction_type="epsilon",
                                    variance_noise=vnoise(si))
            latents = prev.clip(-self.clip_noise, self.clip_noise)
            if trace is not None:
                trace.append((t, latents.clone(), x0.clone()))
        # the reference returns the doubled context; the generator keeps only the conditional half
        # (generator.py:442-448), which is what is returned here.
        self._kv_memo = None
        return self.denormalize(x0), context


class OracleBaseLCM(OracleTwoTower):
    """Base (MSE) LCM: frontend -> causal decoder -> postnet, lcm/models/base_lcm/builder.py:94-167 and
    base_lcm/frontend.py:154-183.  Reuses the attention / FFN / norm blocks of the two-tower oracle."""

    def __init__(self, cfg, sd: Dict[str, Tensor]):
        self.cfg = cfg
        self.sd = sd
        self.model_dim = cfg.model_dim
        self.has_normalizer = "frontend.sonar_normalizer.center" in sd

    def prep_for_denoising(self, opt) -> None:   # the generator calls it for diffusion models only
        pass

    def normalize(self, x: Tensor) -> Tensor:
        if not self.has_normalizer:
            return x
        if getattr(self.cfg, "normalizer_fft", False):
            x = fft_transform(x)
        return (x - self.sd["frontend.sonar_normalizer.center"]) / self.sd["frontend.sonar_normalizer.scale"]

    def denormalize(self, x: Tensor) -> Tensor:
        if not self.has_normalizer:
            return x
        y = x * self.sd["frontend.sonar_normalizer.scale"] + self.sd["frontend.sonar_normalizer.center"]
        return fft_inverse_transform(y) if getattr(self.cfg, "normalizer_fft", False) else y

    def forward(self, x_norm: Tensor, cache: Optional[KVCache] = None) -> Tensor:
        """builder.py:94-124 on already-normalized inputs; `embed_scale` per frontend.py:91, 170."""
        scale = self.cfg.model_dim ** 0.5 if self.cfg.scale_embeddings else 1.0
        h = self.encode(scale * x_norm, cache, frontend="frontend", decoder="lcm", tc=self.cfg.lcm)
        y = linear(h, self.sd, "postnet.fc")                                 # projection.py:80-86
        act = getattr(self.cfg, "postnet_activation", None)
        return y if act is None else ACTIVATIONS[act](y)

    def predict(self, x: Tensor, cache: Optional[KVCache] = None, noise: Optional[Tensor] = None,
                sample: bool = False, temperature: float = 1.0) -> Tensor:
        """predict_next_sentence, builder.py:126-167: normalize, forward, optional N(0, temperature^2) noise on
        the normalized prediction (host-supplied instead of randn_like), denormalize.  All positions."""
        y = self.forward(self.normalize(x), cache)
        if sample and temperature > 0:
            assert noise is not None, "sampling needs host-supplied noise"
            y = y + noise * temperature
        return self.denormalize(y)
