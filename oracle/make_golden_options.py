"""Writes tests/golden/option_statements_golden.pt: the oracle's statements of the layer / projection / normalizer
options that follow fairseq2 code outside the reference tree (post-norm and normformer orders, the stack without its
final norm, the learned residual scale, sinusoidal positions, the two-matrix FFN's PReLU and inner norm, the postnet
activation, the packed real FFT of the "*_fft" normalizers), each on a small seeded problem.

Like tests/golden/toy_two_tower_golden.pt these vectors do not pin the oracle against the reference (which publishes
none and cannot be imported here); they pin the oracle against itself, so that a later change of one of these
restatements shows up in the CPU suite.  Usage: python -m oracle.make_golden_options
"""

import os

import torch

import oracle
from oracle.model import fft_inverse_transform, fft_transform, sinusoidal_positions
from oracle.weights import synthetic_prompts

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "option_statements_golden.pt")

# name -> (overrides of the base arch's tower, overrides of the base arch, init_state_dict arguments)
BASE_CASES = {
    "post_sine": (dict(norm_order="post", pos="sine"), {}, dict(seed=11, qk_gain=2.0, norm_jitter=0.2)),
    "normformer_nofinal_rope_scaled": (dict(norm_order="normformer", final_norm_order="post", pos="rope", scale_residual=True),
                                       {}, dict(seed=12, qk_gain=2.0, norm_jitter=0.2)),
    "relu_ffn_prelu": (dict(use_swiglu=False, activation="prelu"), {}, dict(seed=13, qk_gain=2.0)),
    "postnet_tanh_no_biases": ({}, dict(postnet_activation="tanh", pre_linear_bias=False, postnet_bias=False),
                               dict(seed=14, qk_gain=2.0, w_std=0.05)),
    "fft_normalizer": ({}, dict(normalizer_fft=True), dict(seed=15, qk_gain=2.0)),
}
B, S = 2, 6


def base_case(name):
    tower, arch, init = BASE_CASES[name]
    cfg = oracle.get_base_arch("mini_base_lcm")
    for k, v in tower.items():
        setattr(cfg.lcm, k, v)
    for k, v in arch.items():
        setattr(cfg, k, v)
    sd = oracle.init_state_dict(cfg, **init)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), B, S, seed=200 + init["seed"])
    return cfg, sd, x


def main() -> None:
    blob = {"B": B, "S": S, "base": {}}
    for name in BASE_CASES:
        cfg, sd, x = base_case(name)
        blob["base"][name] = {"prompts": x, "predict": oracle.OracleBaseLCM(cfg, sd).predict(x)}
    g = torch.Generator().manual_seed(31)
    v = torch.randn(3, 1024, generator=g)
    blob["fft"] = {"x": v, "packed": fft_transform(v), "back": fft_inverse_transform(fft_transform(v))}
    blob["sine_rows_3_to_7_dim_256"] = sinusoidal_positions(3, 5, 256)
    torch.save(blob, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
