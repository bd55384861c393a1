"""`DiTTimestepEncoder` as a stand-alone object (reference: lcm/nn/timestep_encoder.py:18-122).

Inside a model the timestep MLP never runs per call: `lcmhip_session_set_schedule` evaluates it once per
distinct timestep into the adaLN table.  This class is for callers that use the encoder by itself: the
sinusoidal frequencies are a host-side table like the scheduler's, the two Linear layers run through
`lcmhip_gemm` (the inner activation in the first product's epilogue).
"""

import math
from typing import Dict, Optional

import torch
from torch import Tensor

from lcm import _hip


class DiTTimestepEncoder:
    def __init__(self, embedding_dim: int, frequency_embedding_size: int = 256, activation_fn_name: str = "silu",
                 device=None, dtype: Optional[torch.dtype] = None) -> None:
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("DiTTimestepEncoder runs only on a GPU (HIP path); there is no CPU fallback")
        self.dtype = dtype or torch.float32
        self.device = device
        self.embedding_dim = embedding_dim
        self.frequency_embedding_size = frequency_embedding_size
        if activation_fn_name is not None and (activation_fn_name not in _hip.ACT or activation_fn_name == "prelu"):
            raise NotImplementedError(f"activation '{activation_fn_name}' is not implemented on the HIP path")
        self.activation_fn_name = activation_fn_name
        chunk = 256 if self.dtype == torch.bfloat16 else 128       # K of a product covers whole 512-byte chunks
        if frequency_embedding_size % chunk or embedding_dim % chunk:
            raise ValueError(f"frequency_embedding_size and embedding_dim must be multiples of {chunk} for {self.dtype}")
        f, d = frequency_embedding_size, embedding_dim
        self._params: Dict[str, Tensor] = {
            "fc1.weight": torch.empty(d, f, device=device, dtype=self.dtype), "fc1.bias": torch.empty(d, device=device, dtype=self.dtype),
            "fc2.weight": torch.empty(d, d, device=device, dtype=self.dtype), "fc2.bias": torch.empty(d, device=device, dtype=self.dtype)}
        self.reset_parameters()

    def reset_parameters(self) -> None:
        """N(0, 0.02) matrices, zero biases (reference :61-70)."""
        for name, t in self._params.items():
            if name.endswith("weight"):
                t.copy_(torch.randn(t.shape, device=t.device, dtype=torch.float32) * 0.02)
            else:
                t.zero_()

    def state_dict(self) -> Dict[str, Tensor]:
        return self._params

    def load_state_dict(self, sd: Dict[str, Tensor]) -> None:
        for k, t in self._params.items():
            t.copy_(sd[k].to(t.dtype))

    @staticmethod
    def sinusoidal_timestep_embedding(timestep: Tensor, frequency_embedding_size: int, max_period: int = 10000) -> Tensor:
        """(N, frequency_embedding_size): cosines of timestep x frequency, then sines (an odd size gets a zero column);
        frequencies exp(-ln(max_period) i / half), fp32 (reference :72-104)."""
        half = frequency_embedding_size // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timestep.device)
        args = timestep[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if frequency_embedding_size % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def _linear(self, x: Tensor, layer: str, epilogue: int) -> Tensor:
        w, b = self._params[layer + ".weight"], self._params[layer + ".bias"]
        out = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=self.dtype)
        _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(x), x.shape[1], _hip.ptr(w), None, _hip.ptr(b), None, None, _hip.ptr(out),
                                        w.shape[0], x.shape[0], w.shape[0], w.shape[1], epilogue, _hip.dtype_code(self.dtype),
                                        None, 0, _hip.stream_ptr(x.device)))
        return out

    @torch.inference_mode()
    def forward(self, timesteps: Tensor) -> Tensor:
        """timesteps of any shape -> (*shape, embedding_dim) in the encoder's dtype (reference :106-122)."""
        shape = tuple(timesteps.shape)
        t = timesteps.reshape(-1).to(self.device)
        freq = self.sinusoidal_timestep_embedding(t, self.frequency_embedding_size).to(self.dtype).contiguous()
        first = _hip.EPI_STORE if self.activation_fn_name is None else (_hip.EPI_RELU | (_hip.ACT[self.activation_fn_name] << 8))
        h = self._linear(freq, "fc1", first)
        return self._linear(h, "fc2", _hip.EPI_STORE).view(*shape, self.embedding_dim)

    __call__ = forward
