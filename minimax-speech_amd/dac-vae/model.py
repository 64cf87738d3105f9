"""Drop-in for dac-vae/model.py: Decoder (:326-379) and DACVAE (:382-506, decode path).

Same constructor arguments and state-dict keys (weight-norm `weight_g` / `weight_v`, Snake `alpha`, the LeakyReLU(0.1)
that model.py:509-514 appends to every Conv1d); `.decode(z)` / `Decoder.forward(x)` run on mmx.dac.DacDecoderEngine.
The encoder half (model.py:146-234,469-483) is SURVEY.md §8f "next" and is not built: `encode` raises.
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
            self._engine = DacDecoderEngine(sd, self.rates, dtype=self.compute_dtype, device=dev,
                                            use_tanh=self.use_tanh_as_final, with_pre=False)
        return self._engine.decode(x.float(), skip_pre=True)


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
        man = shapes.dac_decoder_manifest(latent_dim, decoder_dim, tuple(decoder_rates), d_out)
        register(self, {k: v for k, v in man.items() if k.startswith("de_conv_pre.")})
        self.step = 0

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Reference checkpoints (`checkpoint['generator']`, dac-vae/inference.py:42-46) also hold the encoder and
        en_conv_post; those keys are accepted and ignored (the encoder is not on the hot path)."""
        own = set(self.state_dict().keys())
        sd = {k: v for k, v in state_dict.items() if k in own}
        extra = [k for k in state_dict if k not in own and not k.startswith(("encoder.", "en_conv_post."))]
        if strict and extra:
            raise RuntimeError(f"unexpected keys: {extra[:5]}")
        return super().load_state_dict(sd, strict=strict, **kw)

    def encode(self, audio_data, training=True):
        raise NotImplementedError("DAC-VAE encoder: SURVEY.md §8f 'next' (prompt audio -> latents), not on the hot path yet")

    @torch.inference_mode()
    def decode(self, z: torch.Tensor):
        """z [B, latent_dim, T] -> waveform [B, d_out, T*hop] (model.py:485-488)."""
        from mmx.dac import DacDecoderEngine
        dev = self._device()
        if self._engine is None:
            self._engine = DacDecoderEngine(self.state_dict(), self.decoder_rates, dtype=self.compute_dtype, device=dev,
                                            use_tanh=self.decoder.use_tanh_as_final)
        return self._engine.decode(z.float())
