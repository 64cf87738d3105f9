"""Deterministic synthetic ("random-init") weights for the three hot-path models.

No pretrained weights exist offline, so benchmarks and tests run on random-init weights of the right
architecture (BASELINE.json).  Every tensor is a pure function of (state-dict key, shape, seed) on torch's CPU
generator: the same weights can be regenerated anywhere without shipping 600 M parameters.  Scales are chosen
so activations stay O(1) through the deep stacks (56 transformer blocks, 37 convs); plain random init gives a
~1e-3 waveform and degenerate (uniform) token distributions.
"""
import math
import zlib
from typing import Dict, Tuple

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    return torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)


def _randn(shape, g, std=1.0, mean=0.0):
    return torch.randn(tuple(shape), generator=g) * std + mean


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    g = _gen(seed, name)
    shape = tuple(shape)
    n = name
    # ---- DAC-VAE (weight-normed convs + snake)
    if n.endswith(".alpha"):
        return _randn(shape, g, 0.1, 1.0)
    if n.endswith(".weight_v"):
        return _randn(shape, g)
    if n.endswith(".weight_g"):
        if n.endswith("block.3.0.weight_g"):           # ResidualUnit k1 conv: damp the residual branch
            return torch.full(shape, 0.3)
        if ".block.1.weight_g" in n and n.count("block") == 1:   # ConvTranspose1d, g is per INPUT channel
            return torch.full(shape, 1.0)
        if shape[0] == 1:                              # final conv (C -> 1) ahead of tanh
            return torch.full(shape, 0.25)
        return torch.full(shape, 1.0)
    # ---- norms
    if "norm" in n or n.endswith("block.2.weight") or n.endswith("block.2.bias") or ".out.1." in n \
            or "layernorm" in n:
        if n.endswith("weight"):
            return _randn(shape, g, 0.1, 1.0)
        return _randn(shape, g, 0.05)
    if n.endswith("proj_out.weight"):                  # AttentionBlock output (zero-init in the reference): damped residual
        fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1)
        return _randn(shape, g, 0.5 / math.sqrt(fan_in))
    if n.endswith("pos_bias_u") or n.endswith("pos_bias_v"):
        return _randn(shape, g, 0.2)
    if n.endswith("bias"):
        return _randn(shape, g, 0.05)
    # ---- embeddings
    if n.endswith("embed_tokens.weight") or n.endswith("lm_head.weight"):
        return _randn(shape, g, 0.05)
    if n in ("input_embedding.weight",):
        return _randn(shape, g, 1.0)
    if n in ("speech_embedding.weight", "llm_embedding.weight"):
        return _randn(shape, g, 0.05)
    if n == "llm_decoder.weight":
        return _randn(shape, g, 0.1)                   # logits std ~3: a peaked, non-degenerate nucleus
    # ---- generic Linear / Conv weights: variance preserving, residual output branches damped
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        std = 1.0 / math.sqrt(fan_in)
        if any(t in n for t in ("to_out.0.weight", "ff.net.2.weight", "linear_out.weight", "w_2.weight",
                                "o_proj.weight", "down_proj.weight")):
            std *= 0.5
        if "block.0.weight" in n or "res_conv" in n:   # estimator causal convs feed LayerNorm / Mish
            std *= 1.0
        return _randn(shape, g, std)
    return _randn(shape, g, 0.05)


def synth_state_dict(manifest: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, tuple(v), seed) for k, v in manifest.items()}
