"""oracle/dac.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU fp32 restatement of the DAC-VAE decoder, written functionally
over the reference's state-dict keys.

Follows (reference, read-only):
  dac-vae/model.py:485-488   DACVAE.decode  = de_conv_pre -> Decoder
  dac-vae/model.py:326-379   Decoder        (conv k7, 5 DecoderBlocks, Snake, conv k7, tanh)
  dac-vae/model.py:237-323   DecoderBlock   (Snake, ConvTranspose1d k=2s stride s, 3 ResidualUnits dil 1/3/9)
  dac-vae/model.py:107-143   ResidualUnit   (Snake, conv k7 dil d, Snake, conv k1, + x)
  dac-vae/model.py:509-514   the module-level WNConv1d that SHADOWS layers.WNConv1d:
                             every Conv1d in model.py is followed by LeakyReLU(0.1)
  dac-vae/layers.py:13-14    WNConvTranspose1d (no activation)
  dac-vae/layers.py:18-24    snake(x, a) = x + (a + 1e-9)^-1 * sin(a x)^2
  torch weight_norm (dim=0): w = g * v / ||v||_{dims 1..}
and the encoder side:
  dac-vae/model.py:469-483   DACVAE.encode  = Encoder -> F.leaky_relu(0.01) -> en_conv_post -> (m | logs) -> z
  dac-vae/model.py:195-234   Encoder        (conv k7, 5 EncoderBlocks, Snake, conv k3)
  dac-vae/model.py:146-192   EncoderBlock   (3 ResidualUnits dil 1/3/9 at dim/2, Snake, Conv1d k=2s stride s pad ceil(s/2))
  dac-vae/model.py:457-467   DACVAE.preprocess (right-pad to a hop multiple)
"""
import math

import torch
import torch.nn.functional as F

LRELU = 0.1


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm with dim=0: per-slice-of-dim-0 L2 norm."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * (v / n)


def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    # dac-vae/layers.py:22 (exact operation order)
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def _wn(sd, p):
    return fold_weight_norm(sd[p + ".weight_g"], sd[p + ".weight_v"]), sd.get(p + ".bias")


def wnconv1d_act(sd, p, x, dilation=1, padding=0):
    """model.py:509-514: weight-normed Conv1d followed by LeakyReLU(0.1). `p` is the
    Sequential's prefix; the conv itself is its child '0'."""
    w, b = _wn(sd, p + ".0")
    return F.leaky_relu(F.conv1d(x, w, b, dilation=dilation, padding=padding), LRELU)


def residual_unit(sd, p, x, dilation):
    # model.py:125-143
    pad = ((7 - 1) * dilation) // 2
    y = snake(x, sd[p + ".block.0.alpha"])
    y = wnconv1d_act(sd, p + ".block.1", y, dilation=dilation, padding=pad)
    y = snake(y, sd[p + ".block.2.alpha"])
    y = wnconv1d_act(sd, p + ".block.3", y)
    return x + y


def decoder_block(sd, p, x, stride):
    # model.py:252-284
    x = snake(x, sd[p + ".block.0.alpha"])
    w, b = _wn(sd, p + ".block.1")
    x = F.conv_transpose1d(x, w, b, stride=stride, padding=math.ceil(stride / 2),
                           output_padding=0 if stride % 2 == 0 else 1)
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(sd, f"{p}.block.{2 + i}", x, d)
    return x


def decoder_forward(sd, x, rates, prefix="decoder", use_tanh=True, return_stages=False):
    """Decoder.forward (model.py:373-379). `sd` holds keys '<prefix>.model.N...'."""
    stages = []
    p = prefix + ".model"
    x = wnconv1d_act(sd, p + ".0", x, padding=3)
    stages.append(x)
    for i, s in enumerate(rates):
        x = decoder_block(sd, f"{p}.{1 + i}", x, s)
        stages.append(x)
    n = len(rates)
    x = snake(x, sd[f"{p}.{n + 1}.alpha"])
    x = wnconv1d_act(sd, f"{p}.{n + 2}", x, padding=3)
    x = torch.tanh(x) if use_tanh else torch.clamp(x, -1.0, 1.0)
    return (x, stages) if return_stages else x


def decode(sd, z, rates, use_tanh=True):
    """DACVAE.decode (model.py:485-488): z [B, D_lat, T] -> waveform [B, d_out, T*hop]."""
    z = wnconv1d_act(sd, "de_conv_pre", z)
    return decoder_forward(sd, z, rates, "decoder", use_tanh)


def wnconv1d_strided_act(sd, p, x, stride, padding):
    w, b = _wn(sd, p + ".0")
    return F.leaky_relu(F.conv1d(x, w, b, stride=stride, padding=padding), LRELU)


def encoder_block(sd, p, x, stride):
    # model.py:159-192
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(sd, f"{p}.block.{i}", x, d)
    x = snake(x, sd[p + ".block.3.alpha"])
    return wnconv1d_strided_act(sd, p + ".block.4", x, stride, math.ceil(stride / 2))


def encoder_forward(sd, x, rates, prefix="encoder", return_stages=False):
    """Encoder.forward (model.py:233-234). `sd` holds keys '<prefix>.block.N...'; x [B, d_in, T]."""
    stages = []
    p = prefix + ".block"
    x = wnconv1d_act(sd, p + ".0", x, padding=3)
    stages.append(x)
    for i, s in enumerate(rates):
        x = encoder_block(sd, f"{p}.{1 + i}", x, s)
        stages.append(x)
    n = len(rates)
    x = snake(x, sd[f"{p}.{n + 1}.alpha"])
    x = wnconv1d_act(sd, f"{p}.{n + 2}", x, padding=1)
    return (x, stages) if return_stages else x


def preprocess(audio, hop):
    """DACVAE.preprocess (model.py:457-467): zero right-pad to a multiple of the hop length."""
    length = audio.shape[-1]
    return F.pad(audio, (0, math.ceil(length / hop) * hop - length))


def encode(sd, audio, rates, noise):
    """DACVAE.encode (model.py:469-483): audio [B, d_in, T] -> (z, m, logs) each [B, D_lat, T/hop].
    `noise` stands for the reference's torch.randn_like(m) draw."""
    x = encoder_forward(sd, audio, rates, "encoder")
    x = F.leaky_relu(x)                                   # default slope 0.01 (model.py:475)
    x = wnconv1d_act(sd, "en_conv_post", x)
    lat = x.shape[1] // 2
    m, logs = torch.split(x, lat, dim=1)
    logs = torch.clamp(logs, min=-14.0, max=14.0)
    z = m + noise * torch.exp(logs)
    return z, m, logs
