"""The oracle's arithmetic with activations rounded to bf16 exactly where the HIP kernels round them
(TEST INFRASTRUCTURE, like the rest of this package).

Why: a bf16 run differs from the fp32 oracle for two reasons -- the inherent rounding of bf16 operands, and any
defect of a kernel.  On a well-conditioned network both are small; a random 13-layer adaLN network amplifies
the first so much that the fp32 oracle can no longer tell them apart.  This module keeps the oracle's algorithm
(same reference citations as oracle/model.py) in fp32 arithmetic but rounds to bf16 at the HIP path's storage
points, so that what is left between it and the HIP output is kernel error (plus the rare rounding flip caused
by a different fp32 summation order).

What it can and cannot show.  Rounding flips do not stay rare: one activation that rounds the other way shifts
every later fp32 value a little, which flips more roundings, and after a few layers two runs that differ ONLY in
summation order disagree in most elements by a bf16 step or two.  So the emulation does not track a bf16 kernel
run bit for bit through a deep network; what it gives is the size of the inherent bf16 noise on a given network
(emulation vs the fp32 oracle) and the drift between two correct bf16 implementations (`summation_order`: the
same emulation with split-K sums).  A kernel run is indistinguishable from a correct implementation when its
distance to the emulation is of the size of that drift, and its distance to the fp32 oracle of the size of the
emulation's.

Rounding points mirrored (bf16 models of liblcm_hip.so):
  * weights, biases, normalizer buffers: bf16 (the caller's state dict is rounded once here);
  * normalizer output, every norm(+adaLN) output, q/k/v projections (after bias), RoPE outputs, attention
    outputs, SwiGLU / ReLU activations, the contextualizer output, cross-attention K/V: rounded once when stored;
  * the residual stream, latents, DDIM / guidance arithmetic, softmax statistics, adaLN table entries: fp32;
  * the denoiser's self-attention (one position) uses the folded W_ov = bf16(W_o W_v), b_ov = bf16(W_o b_v + b_o);
  * causal prefill blocks of >= 4 positions run the matrix-core attention kernel, which rounds the softmax
    numerators to bf16 per 32-key tile (online softmax) before the P V product; single-query attention keeps
    them in fp32;
  * the timestep embedding chain: bf16 inputs to fc1 / fc2 / modulator, fp32 outputs, embedding rounded to bf16;
  * generated embeddings are stored in bf16 by the generator (seqs buffer in the model dtype).
"""

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .archs import TowerConfig
from .model import ACTIVATIONS, KVCache, OracleBaseLCM, OracleTwoTower, causal_mask, rope

PREFILL_MIN = 4      # shortest causal query block that takes the matrix-core prefill kernel
PREFILL_TILE = 32    # keys per tile of that kernel


def r16(x: Tensor) -> Tensor:
    """Round to bf16 (nearest even) and back to fp32."""
    return x.bfloat16().float()


def _silu(x: Tensor) -> Tensor:
    return x / (1.0 + torch.exp(-x))


_K_SPLITS = 1


class summation_order:
    """Context manager: every product inside it is summed as `splits` partial products over ranges of K (what a
    split-K kernel does).  Same arithmetic, different fp32 summation order: a second emulation run under it shows
    how far two CORRECT implementations of the same rounding points drift apart on a given network."""

    def __init__(self, splits: int):
        self.splits = splits

    def __enter__(self):
        global _K_SPLITS
        self.prev, _K_SPLITS = _K_SPLITS, self.splits
        return self

    def __exit__(self, *exc):
        global _K_SPLITS
        _K_SPLITS = self.prev
        return False


def _matmul_t(x: Tensor, w: Tensor) -> Tensor:
    """x @ w.T in fp32, optionally as a sum of K-range partial products."""
    K = w.shape[1]
    if _K_SPLITS <= 1 or K % _K_SPLITS:
        return F.linear(x, w)
    step = K // _K_SPLITS
    acc = F.linear(x[..., :step], w[:, :step])
    for k0 in range(step, K, step):
        acc = acc + F.linear(x[..., k0:k0 + step], w[:, k0:k0 + step])
    return acc


def _lin(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """bf16 operands, fp32 accumulation, fp32 result (+ bias)."""
    y = _matmul_t(x, sd[prefix + ".weight"])
    b = sd.get(prefix + ".bias")
    return y if b is None else y + b


def _norm32(x: Tensor, sd: Dict[str, Tensor], prefix: str, kind: str, eps: float,
            shift: Optional[Tensor] = None, scale: Optional[Tensor] = None) -> Tensor:
    """Norm (+ adaLN modulation) entirely in fp32, rounded once (k_norm_modulate / k_norm_pending)."""
    w = sd.get(prefix + ".weight")          # absent for the "unit" style
    if kind == "rms":
        y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w
    else:                                   # "standard" (weight + bias), "fp32" (weight), "unit" (neither)
        mean = x.mean(-1, keepdim=True)
        var = (x - mean).pow(2).mean(-1, keepdim=True)
        y = (x - mean) * torch.rsqrt(var + eps)
        if w is not None:
            y = y * w
        if kind == "standard":
            y = y + sd[prefix + ".bias"]
    if scale is not None:
        y = y * (1.0 + scale) + shift
    return r16(y)


def _attend_single(q: Tensor, k: Tensor, v: Tensor, key_lens: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """k_attention: fp32 scores, softmax and P V; q (N,H,S,hd) already rounded, output rounded."""
    hd = q.shape[-1]
    w = torch.matmul(q * (hd ** -0.5), k.transpose(-1, -2))
    if mask is not None:
        w = w + mask
    if key_lens is not None:
        pad = torch.arange(k.shape[-2])[None, :] >= key_lens[:, None]
        w = w.masked_fill(pad[:, None, None, :], float("-inf"))
    return r16(torch.matmul(torch.softmax(w, dim=-1), v))


def _attend_prefill(q: Tensor, k: Tensor, v: Tensor, base_keys: int) -> Tensor:
    """k_attention_prefill: query j sees keys [0, base_keys + j + 1); online softmax over 32-key tiles with the
    numerators rounded to bf16 for the P V product while the denominators stay fp32."""
    N, H, S, hd = q.shape
    K = k.shape[-2]
    scale = hd ** -0.5
    vis = torch.arange(K)[None, :] < (base_keys + torch.arange(S)[:, None] + 1)       # (S, K)
    m = torch.full((N, H, S, 1), float("-inf"))
    l = torch.zeros(N, H, S, 1)
    o = torch.zeros(N, H, S, hd)
    for k0 in range(0, K, PREFILL_TILE):
        k1 = min(K, k0 + PREFILL_TILE)
        live = vis[:, k0:k1].any(dim=1)                                                  # queries that see this tile
        if not bool(live.any()):
            break
        s = torch.matmul(q, k[:, :, k0:k1].transpose(-1, -2)) * scale
        s = s.masked_fill(~vis[None, None, :, k0:k1], float("-inf"))
        mn = torch.maximum(m, s.amax(dim=-1, keepdim=True))
        mn = torch.where(live[None, None, :, None], mn, m)                               # untouched rows keep their state
        alpha = torch.where(torch.isinf(m) & torch.isinf(mn), torch.ones_like(m), torch.exp(m - mn))
        p = torch.exp(s - mn)
        p = torch.where(live[None, None, :, None], p, torch.zeros_like(p))
        l = l * alpha + p.sum(dim=-1, keepdim=True)
        o = o * alpha + torch.matmul(r16(p), v[:, :, k0:k1])
        m = mn
    return r16(o / l)


class _Bf16Blocks:
    """Attention / FFN blocks with the HIP path's rounding points; mixed into the two models below."""

    sd: Dict[str, Tensor]

    def _ctx_mha(self, prefix: str, tc: TowerConfig, h: Tensor, cache: Optional[KVCache], layer: int, start: int) -> Tensor:
        sd, H = self.sd, tc.num_heads
        S = h.shape[1]
        q = r16(_lin(h, sd, prefix + ".q_proj")).unflatten(-1, (H, -1)).transpose(1, 2)
        k = r16(_lin(h, sd, prefix + ".k_proj")).unflatten(-1, (H, -1)).transpose(1, 2)
        v = r16(_lin(h, sd, prefix + ".v_proj")).unflatten(-1, (H, -1)).transpose(1, 2)
        if tc.pos == "rope":
            q, k = r16(rope(q, start, tc.rope_theta)), r16(rope(k, start, tc.rope_theta))
        if cache is not None:
            k, v = cache.append(layer, k, v)
        if S >= PREFILL_MIN:
            o = _attend_prefill(q, k, v, start)
        else:
            o = _attend_single(q, k, v, None, causal_mask(S, start + S))
        return _lin(o.transpose(1, 2).flatten(2, 3), sd, prefix + ".output_proj")       # fp32, added to the residual

    def _ffn16(self, prefix: str, tc: TowerConfig, h: Tensor) -> Tensor:
        sd = self.sd
        if tc.use_swiglu:
            act = r16(_lin(h, sd, prefix + ".inner_proj") * _silu(_lin(h, sd, prefix + ".gate_proj")))
        else:
            act = r16(ACTIVATIONS[tc.activation](_lin(h, sd, prefix + ".inner_proj")))
        return _lin(act, sd, prefix + ".output_proj")

    def _stack(self, x_norm16: Tensor, cache: Optional[KVCache], frontend: str, decoder: str, tc: TowerConfig,
               embed_scale: float = 1.0) -> Tensor:
        """Frontend + causal decoder stack; input already normalized and rounded; returns the rounded output."""
        cfg, sd = self.cfg, self.sd
        S = x_norm16.shape[1]
        start = cache.step_nr if cache is not None else 0
        x = _lin(x_norm16, sd, frontend + ".pre_linear")
        if tc.pos == "learned":
            x = x + sd[frontend + ".pos_encoder.weight"][start:start + S]
        for i in range(tc.num_layers):
            p = f"{decoder}.layers.{i}"
            h = _norm32(x, sd, p + ".self_attn_layer_norm", tc.norm, cfg.norm_eps)
            x = x + self._ctx_mha(p + ".self_attn", tc, h, cache, i, start)
            h = _norm32(x, sd, p + ".ffn_layer_norm", tc.norm, cfg.norm_eps)
            x = x + self._ffn16(p + ".ffn", tc, h)
        return _norm32(x, sd, decoder + ".layer_norm", tc.norm, cfg.norm_eps)


class OracleTwoTowerBf16(_Bf16Blocks, OracleTwoTower):
    def __init__(self, cfg, sd: Dict[str, Tensor]):
        OracleTwoTower.__init__(self, cfg, {k: r16(v.float()) for k, v in sd.items()})
        assert not (cfg.context.qk_layernorm or cfg.denoiser.qk_layernorm), "the bf16 emulation does not model q / k LayerNorm"
        sd = self.sd
        self.ov: List[Tuple[Tensor, Tensor]] = []
        self.uconst: List[Tensor] = []
        for i in range(cfg.denoiser.num_layers):
            p = f"denoiser.layers.{i}"
            wo, wv = sd[p + ".self_attn.output_proj.weight"], sd[p + ".self_attn.v_proj.weight"]
            bo = sd.get(p + ".self_attn.output_proj.bias")
            b_ov = wo @ sd[p + ".self_attn.v_proj.bias"]
            self.ov.append((r16(wo @ wv), r16(b_ov + bo if bo is not None else b_ov)))
            u = sd[p + ".cross_attention.output_proj.weight"] @ sd[p + ".cross_attention.v_proj.bias"]
            bo = sd.get(p + ".cross_attention.output_proj.bias")
            self.uconst.append(u + bo if bo is not None else u)                       # fp32 (STORE_F32)

    # -- normalizer as fused into the staging kernel: fp32 in, bf16 out
    def normalize(self, x: Tensor) -> Tensor:
        return r16((x.float() - self.sd["sonar_normalizer.center"]) / self.sd["sonar_normalizer.scale"])

    def encode(self, x_norm: Tensor, cache: Optional[KVCache] = None, **kw) -> Tensor:
        return self._stack(r16(x_norm), cache, "encoder_frontend", "context_encoder", self.cfg.context)

    def _mod_table(self, timesteps: List[int]) -> Tensor:
        """[S][L][9d] fp32: timestep_encoder.py:72-122 + lcm_denoiser.py:60-64 through bf16 GEMM operands."""
        sd, cfg = self.sd, self.cfg
        half = cfg.timestep_freq_dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        args = torch.tensor(timesteps, dtype=torch.float32)[:, None] * freqs[None]
        emb = r16(torch.cat([torch.cos(args), torch.sin(args)], dim=-1))
        h = r16(_silu(_lin(emb, sd, "denoiser.embed_time.fc1")))
        e = r16(_lin(h, sd, "denoiser.embed_time.fc2"))
        a = r16(_silu(e))
        return torch.stack([_lin(a, sd, f"denoiser.layers.{i}.modulator.fc") for i in range(cfg.denoiser.num_layers)], dim=1)

    def prep_for_denoising(self, opt) -> None:
        OracleTwoTower.prep_for_denoising(self, opt)
        self.mod_table = self._mod_table(self.scheduler.timesteps.tolist())

    def _denoise_rows(self, lat16: Tensor, step_index: int, kv: List[Tuple[Tensor, Tensor]], cfg_rows: bool) -> Tensor:
        """One denoiser evaluation of the sampling loop: lat16 (B,D) rounded latents; returns the model output of
        the conditional rows and, with `cfg_rows`, of the unconditional rows: (B or 2B, D) fp32."""
        cfg, sd, tc = self.cfg, self.sd, self.cfg.denoiser
        B, d, H = lat16.shape[0], cfg.model_dim, tc.num_heads
        x = _lin(lat16, sd, "denoiser.initial_proj.fc")
        if cfg_rows:
            x = torch.cat([x, x], dim=0)
        for i in range(tc.num_layers):
            p = f"denoiser.layers.{i}"
            mod = self.mod_table[step_index, i]
            sh = [mod[j * d:(j + 1) * d] for j in range(9)]
            h = _norm32(x, sd, p + ".self_attn_layer_norm", tc.norm, cfg.norm_eps, sh[0], sh[1])
            x = x + sh[2] * (_matmul_t(h, self.ov[i][0]) + self.ov[i][1])
            hc = _norm32(x[:B], sd, p + ".cross_attention_layer_norm", tc.norm, cfg.norm_eps, sh[3], sh[4])
            q = r16(_lin(hc, sd, p + ".cross_attention.q_proj")).unflatten(-1, (H, -1)).unsqueeze(2)     # (B,H,1,hd)
            o = _attend_single(q, kv[i][0], kv[i][1], None, None).squeeze(2).flatten(1)
            xc = x[:B] + sh[5] * _lin(o, sd, p + ".cross_attention.output_proj")
            x = torch.cat([xc, x[B:] + sh[5] * self.uconst[i]], dim=0) if cfg_rows else xc
            h = _norm32(x, sd, p + ".ffn_layer_norm", tc.norm, cfg.norm_eps, sh[6], sh[7])
            x = x + sh[8] * self._ffn16(p + ".ffn", tc, h)
        h = _norm32(x, sd, "denoiser.layer_norm", tc.norm, cfg.norm_eps)
        return _lin(h, sd, "denoiser.final_proj.fc")

    def _cross_kv(self, context16: Tensor) -> List[Tuple[Tensor, Tensor]]:
        """Cross-attention cache per denoiser layer: position 0 = the k/v biases (the prepended zero vector),
        then the rounded projections of the context (lcm_denoiser.py:193-199, 505-513)."""
        sd, tc = self.sd, self.cfg.denoiser
        H = tc.num_heads
        out = []
        for i in range(tc.num_layers):
            p = f"denoiser.layers.{i}.cross_attention"
            kv = []
            for proj in ("k_proj", "v_proj"):
                b = sd[p + f".{proj}.bias"]
                t = torch.cat([b.expand(context16.shape[0], 1, -1), r16(_lin(context16, sd, p + "." + proj))], dim=1)
                kv.append(t.unflatten(-1, (H, -1)).transpose(1, 2))
            out.append((kv[0], kv[1]))
        return out

    def predict_next_sentence(self, x: Tensor, context: Tensor, cache: KVCache, noise: Tensor,
                              trace: Optional[List] = None, variance_noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """builder.py:262-461 with the HIP path's data flow (cached cross K/V, constant unconditional cross-attention,
        folded self-attention); the generated embedding is rounded like the generator's bf16 sequence buffer."""
        sch = self.scheduler
        new_ctx = self.encode(self.normalize(x), cache)
        cache.increment_step_nr(1)
        context = torch.cat((context, new_ctx), dim=1)
        kv = self._cross_kv(context)
        latents = self.initial_latents(noise).squeeze(1)
        x0 = latents
        for si, t in enumerate(sch.timesteps.tolist()):
            guided = self.do_cfg and t != sch.num_train_steps - 1
            vn = None if variance_noise is None else variance_noise[si].reshape(latents.shape)
            out = self._denoise_rows(r16(latents), si, kv, cfg_rows=guided)
            if guided:
                inp = torch.cat([latents, latents], dim=0)
                eps = self.apply_cfg(sch.get_epsilon(out, inp, t))
                prev, x0 = sch.step(eps, t, latents, eta=self.ddim_eta, epsilon_scaling=self.epsilon_scaling,
                                    prediction_type="epsilon", variance_noise=vn)
            else:
                prev, x0 = sch.step(out, t, latents, eta=self.ddim_eta, epsilon_scaling=self.epsilon_scaling,
                                    variance_noise=vn)
            latents = prev.clip(-self.clip_noise, self.clip_noise)
            if trace is not None:
                trace.append((t, latents.clone(), x0.clone()))
        return r16(self.denormalize(x0)).unsqueeze(1), context


class OracleBaseLCMBf16(_Bf16Blocks, OracleBaseLCM):
    def __init__(self, cfg, sd: Dict[str, Tensor]):
        OracleBaseLCM.__init__(self, cfg, {k: r16(v.float()) for k, v in sd.items()})
        assert not cfg.lcm.qk_layernorm, "the bf16 emulation does not model q / k LayerNorm"

    def normalize(self, x: Tensor) -> Tensor:
        if not self.has_normalizer:
            return r16(x.float())
        return r16((x.float() - self.sd["frontend.sonar_normalizer.center"]) / self.sd["frontend.sonar_normalizer.scale"])

    def forward(self, x_norm: Tensor, cache: Optional[KVCache] = None) -> Tensor:
        scale = self.cfg.model_dim ** 0.5 if self.cfg.scale_embeddings else 1.0
        h = self._stack(r16(scale * x_norm), cache, "frontend", "lcm", self.cfg.lcm)
        return _lin(h, self.sd, "postnet.fc")

    def predict(self, x: Tensor, cache: Optional[KVCache] = None, noise: Optional[Tensor] = None,
                sample: bool = False, temperature: float = 1.0, round_output: bool = True) -> Tensor:
        """The library returns fp32 predictions; the generator stores them in its bf16 sequence buffer
        (`round_output`), a plain forward pass does not."""
        y = OracleBaseLCM.predict(self, x, cache, noise, sample, temperature)
        return r16(y) if round_output else y
