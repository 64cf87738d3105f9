"""Deterministic synthetic ("random-init") weights for the three hot-path models.

No pretrained weights exist offline, so benchmarks and tests run on random-init weights of the right
architecture (BASELINE.json).  Every tensor is a pure function of (state-dict key, shape, seed) on torch's CPU
generator: the same weights can be regenerated anywhere without shipping 600 M parameters.  Scales are chosen
so activations stay O(1) through the deep stacks (56 transformer blocks, 37 convs); plain random init gives a
~1e-3 waveform and degenerate (uniform) token distributions.

Two kinds of checkpoint (`kind` of synth_state_dict):

  * "bf16" (default): every matrix-shaped parameter (Linear / Conv / embedding weights) is **bf16-representable** — a
    checkpoint that ships in bf16 (Qwen2.5-0.5B does).  Both sides hold the weights exactly: the CPU oracle computes in
    fp32 on these values, the GPU streams them as bf16 with no rounding.  Weight-normed convs (DAC-VAE) are in the state
    torch's weight_norm leaves a freshly initialised module in: weight_g = ||weight_v|| per output channel, so the
    effective weight g * v / ||v|| is weight_v itself.
  * "fp32": what the reference's own loaders hand over (speech/cosyvoice/cli/model.py:67-75 loads fp32 llm.pt / flow.pt,
    dac-vae/inference.py:42-46 a generator with trained weight_g / weight_v): the same draws WITHOUT the rounding to bf16,
    and weight_g = s * ||weight_v|| with a per-channel s in [0.6, 1.5] that is not a power of two (a trained weight norm:
    dac-vae/model.py:509-514, layers.py:9-14), so the folded conv weight g * v / ||v|| is a general fp32 value and a
    wrong norm axis (the ConvTranspose1d's weight_g is [Cin, 1, 1]) changes the result.

Vectors (biases, norm gains, snake alphas) stay fp32 in both.
"""
import math
import zlib
from typing import Dict, Tuple

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    return torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)


def _randn(shape, g, std=1.0, mean=0.0):
    return torch.randn(tuple(shape), generator=g) * std + mean


def _bf16_exact(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


# per-output-channel gain the weight-normed conv ends up with (effective weight = weight_v, see the module docstring)
def _wn_gain(n: str, shape) -> float:
    if n.endswith("block.3.0.weight_v"):               # ResidualUnit k1 conv: damp the residual branch
        return 0.3
    if shape[0] == 1:                                  # final conv (C -> 1) ahead of tanh
        return 0.25
    return 1.0


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int, kind: str = "bf16") -> torch.Tensor:
    t = _synth_tensor(name, shape, seed)
    if kind == "bf16" and len(shape) >= 2 and not name.endswith(".alpha") and not name.endswith(".weight_g"):
        t = _bf16_exact(t)
    return t


def wn_scale(name: str, n: int, seed: int) -> torch.Tensor:
    """Per-channel ratio weight_g / ||weight_v|| of the "fp32" kind: uniform in [0.6, 1.5], never a power of two."""
    s = 0.6 + 0.9 * torch.rand(n, generator=_gen(seed, name + "#g"))
    return torch.where((s - 1.0).abs() < 1e-3, s + 0.01, s)


def _synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    g = _gen(seed, name)
    shape = tuple(shape)
    n = name
    # ---- DAC-VAE (weight-normed convs + snake)
    if n.endswith(".alpha"):
        return _randn(shape, g, 0.1, 1.0)
    if n.endswith(".weight_v"):
        fan = 1
        for s_ in shape[1:]:
            fan *= s_
        return _randn(shape, g, _wn_gain(n, shape) / math.sqrt(fan))
    if n.endswith(".weight_g"):
        raise KeyError("weight_g is derived from weight_v: use synth_state_dict")
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


def synth_state_dict(manifest: Dict[str, Tuple[int, ...]], seed: int = 0, kind: str = "bf16") -> Dict[str, torch.Tensor]:
    assert kind in ("bf16", "fp32")
    sd = {k: synth_tensor(k, tuple(v), seed, kind) for k, v in manifest.items() if not k.endswith(".weight_g")}
    for k, shp in manifest.items():
        if k.endswith(".weight_g"):
            # torch.nn.utils.weight_norm at init: g = ||v|| over every dim but 0 (the same call the module's forward makes)
            v = sd[k[:-1] + "v"]
            g = torch.norm_except_dim(v, 2, 0).reshape(tuple(shp)).clone()
            if kind == "fp32":
                g = g * wn_scale(k, g.numel(), seed).reshape(g.shape)
            sd[k] = g
    return {k: sd[k] for k in manifest}
