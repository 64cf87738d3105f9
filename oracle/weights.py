"""oracle/weights.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Deterministic synthetic weights.  No pretrained weights exist offline, and the
full-size state dicts (35 M / 113 M / 494 M parameters) cannot be committed as fixtures.  Instead every
tensor is a pure function of (its state-dict key, its shape, a seed) on torch's CPU generator, so

  * oracle/gen_golden.py loads these weights INTO the reference's own modules, runs them, and commits
    only inputs + outputs (+ the key->shape manifest) under tests/golden/;
  * tests and bench regenerate the identical weights anywhere (same torch build) without the reference.

The scales are chosen so activations stay O(1) through the deep stacks (56 transformer blocks, 37
convs): random-init reference weights give a ~1e-3 waveform, which would make the 1e-3 abs parity
tolerance of BASELINE.json vacuous.
"""
import json
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


def load_manifest(path: str) -> Dict[str, Tuple[int, ...]]:
    with open(path) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def save_manifest(sd, path: str):
    with open(path, "w") as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)


def sampler_cases(n: int = 200, seed: int = 31):
    """Inputs of the sampler golden cases (tests/golden/sampler.npz holds the reference's outputs; the
    torch seed of case s is 1000 + s): log-prob vectors of four peakedness levels and decoded-token
    histories of length 0..12, every third one ending in the arg-max token (forces the repetition branch)."""
    g = torch.Generator().manual_seed(seed)
    cases = []
    for s in range(n):
        scale = [0.5, 2.0, 4.0, 8.0][s % 4]
        logp = (torch.randn(6564, generator=g) * scale).log_softmax(0)
        hist = torch.randint(0, 6561, (s % 13,), generator=g).tolist()
        if s % 3 == 0 and hist:
            hist[-1] = int(logp.argmax())
        cases.append((logp, hist))
    return cases
