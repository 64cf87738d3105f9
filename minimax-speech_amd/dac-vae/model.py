"""Drop-in for dac-vae/model.py: Encoder (:195-234), Decoder (:326-379) and DACVAE (:382-506).

Same constructor arguments and state-dict keys (weight-norm `weight_g` / `weight_v`, Snake `alpha`, the LeakyReLU(0.1)
that model.py:509-514 appends to every Conv1d); `.decode(z)` / `Decoder.forward(x)` run on mmx.dac.DacDecoderEngine,
`.encode(audio)` / `.forward(audio)` on mmx.dac.DacEncoderEngine.  Inference only (no autograd, no training step).
"""
import math
from typing import List

import numpy as np
import torch
from torch import nn

import _mmx_path  # noqa: F401
from mmx import shapes
from mmx.shell import EngineHost, register


class Decoder(EngineHost):
    def __init__(self, input_channel, channels, rates, d_out: int = 1, norm: bool = False, activation: str = "snake",
                 alpha: float = 1.0, scale_residual: bool = False, use_tanh_as_final: bool = True,
                 use_bias_at_final: bool = True):
        super().__init__()
        if norm or activation != "snake" or scale_residual or not use_bias_at_final or d_out != 1:
            raise NotImplementedError("only the configuration of dac-vae/configs/configx2.yml is on the hot path")
        self.input_channel, self.channels, self.rates = input_channel, channels, list(rates)
        self.use_tanh_as_final = use_tanh_as_final
        man = shapes.dac_decoder_manifest(input_channel, channels, tuple(rates), d_out)
        register(self, man, prefix="decoder.")            # -> self.model.N...

    @torch.inference_mode()
    def forward(self, x):
        """x [B, C_in, T] (the output of de_conv_pre) -> [B, 1, T*hop]."""
        from mmx.dac import DacDecoderEngine
        dev = self._device()
        if self._engine is None:
            sd = {"decoder." + k: v for k, v in self.state_dict().items()}
            self._engine = DacDecoderEngine(sd, self.rates, dtype=self.compute_dtype, device=dev, wplanes=getattr(self, 'weight_planes', False),
                                            use_tanh=self.use_tanh_as_final, with_pre=False)
        return self._engine.decode(x.float(), skip_pre=True)


class Encoder(EngineHost):
    """Parameter shell for model.py:195-234; the arithmetic runs inside DACVAE.encode (it needs en_conv_post too)."""

    def __init__(self, d_model: int = 64, strides: list = [2, 4, 8, 8], d_latent: int = 64, d_in: int = 1,
                 activation: str = "snake", alpha: float = 1.0, scale_residual: bool = False):
        super().__init__()
        if activation != "snake" or scale_residual or d_in != 1:
            raise NotImplementedError("only the configuration of dac-vae/configs/configx2.yml is on the hot path")
        self.strides = list(strides)
        man = shapes.dac_encoder_manifest(d_latent, d_model, tuple(strides), d_in)
        register(self, {k: v for k, v in man.items() if k.startswith("encoder.")}, prefix="encoder.")
        self.enc_dim = d_model * 2 ** len(strides)


class DACVAE(EngineHost):
    def __init__(self, encoder_dim: int = 64, encoder_rates: List[int] = [2, 4, 8, 8], latent_dim: int = 64,
                 decoder_dim: int = 1536, decoder_rates: List[int] = [8, 8, 4, 2], sample_rate: int = 44100,
                 d_in: int = 2, d_out: int = 2, weight_init: str = "xavier", norm: bool = False,
                 activation: str = "snake", alpha: float = 1.0, gain: float = 0.02, scale_residual: bool = False,
                 use_tanh_as_final: bool = True, use_bias_at_final: bool = True):
        super().__init__()
        self.encoder_dim, self.encoder_rates = encoder_dim, encoder_rates
        self.decoder_dim, self.decoder_rates = decoder_dim, decoder_rates
        self.sample_rate, self.d_in, self.d_out = sample_rate, d_in, d_out
        if latent_dim is None:
            latent_dim = encoder_dim * (2 ** len(encoder_rates))
        self.latent_dim = latent_dim
        self.hop_length = int(np.prod(encoder_rates))
        self.decoder = Decoder(latent_dim, decoder_dim, decoder_rates, d_out=d_out, norm=norm, activation=activation,
                               alpha=alpha, scale_residual=scale_residual, use_tanh_as_final=use_tanh_as_final,
                               use_bias_at_final=use_bias_at_final)
        self.encoder = Encoder(encoder_dim, encoder_rates, latent_dim, d_in=d_in, activation=activation, alpha=alpha,
                               scale_residual=scale_residual)
        man = shapes.dac_decoder_manifest(latent_dim, decoder_dim, tuple(decoder_rates), d_out)
        register(self, {k: v for k, v in man.items() if k.startswith("de_conv_pre.")})
        eman = shapes.dac_encoder_manifest(latent_dim, encoder_dim, tuple(encoder_rates), d_in)
        register(self, {k: v for k, v in eman.items() if k.startswith("en_conv_post.")})
        self.step = 0
        self._enc_engine = None
        self.noise_generator = None        # optional torch.Generator (cuda) for the VAE draw; default: global cuda RNG

    def _invalidate(self):
        super()._invalidate()
        self._enc_engine = None

    def preprocess(self, audio_data, sample_rate):
        """model.py:457-467: zero right-pad to a multiple of the hop length."""
        if sample_rate is None:
            sample_rate = self.sample_rate
        assert sample_rate == self.sample_rate
        length = audio_data.shape[-1]
        right_pad = math.ceil(length / self.hop_length) * self.hop_length - length
        return nn.functional.pad(audio_data, (0, right_pad))

    @torch.inference_mode()
    def encode(self, audio_data: torch.Tensor, training: bool = True, noise: torch.Tensor = None):
        """audio [B, 1, T] (T a hop multiple, see preprocess) -> (z, m, logs), each [B, latent_dim, T/hop]
        (model.py:469-483).  `noise` optionally supplies the randn_like(m) draw (parity tests)."""
        from mmx.dac import DacEncoderEngine
        dev = self._device()
        if self._enc_engine is None:
            self._enc_engine = DacEncoderEngine(self.state_dict(), self.encoder_rates, dtype=self.compute_dtype, device=dev)
        return self._enc_engine.encode(audio_data.float(), noise=noise, generator=self.noise_generator)

    @torch.inference_mode()
    def forward(self, audio_data: torch.Tensor, sample_rate: int = 24000):
        """model.py:490-506: preprocess -> encode -> decode, cropped back to the input length."""
        length = audio_data.shape[-1]
        audio_data = self.preprocess(audio_data, sample_rate)
        z, m, logs = self.encode(audio_data)
        x = self.decode(z)
        return {"audio": x[..., :length], "z": z, "mu": m, "logs": logs}

    @torch.inference_mode()
    def decode(self, z: torch.Tensor):
        """z [B, latent_dim, T] -> waveform [B, d_out, T*hop] (model.py:485-488)."""
        from mmx.dac import DacDecoderEngine
        dev = self._device()
        if self._engine is None:
            self._engine = DacDecoderEngine(self.state_dict(), self.decoder_rates, dtype=self.compute_dtype, device=dev, wplanes=getattr(self, 'weight_planes', False),
                                            use_tanh=self.decoder.use_tanh_as_final)
        return self._engine.decode(z.float())
