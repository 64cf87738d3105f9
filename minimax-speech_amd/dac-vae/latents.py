"""The `*_latent2x.pt` latent file format that connects DAC-VAE to the TTS flow model.

Writer side follows dac-vae/extract_dac_latents.py:20-46 (clamp to [-1, 1], encode UNPADDED audio) and :171-196
(the saved dict and the `<audio stem>_latent2x.pt` naming); reader side follows
speech/cosyvoice/dataset/processor.py:149-159 (`z` -> [T, D], trimmed so latents = token_latent_ratio * tokens).
The encoder itself runs on the HIP path (model.DACVAE.encode); file IO is torch.save / torch.load as in the reference,
so files written by either side load in the other.  Audio decoding / resampling (librosa in the reference) and the
multi-process driver of that script are outside the hot path: callers hand over a mono float waveform.
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

LATENT_SUFFIX = "_latent2x.pt"


def latent_path(audio_path: str) -> str:
    """a/b/c/d.wav -> a/b/c/d_latent2x.pt (extract_dac_latents.py:166-168)."""
    return os.path.splitext(audio_path)[0] + LATENT_SUFFIX


@torch.inference_mode()
def latent_record(model, audio, sample_rate: int, original_path: str = "") -> Dict:
    """Encodes one mono waveform (numpy or torch, [T]) and returns the reference's latent dict (CPU tensors)."""
    if sample_rate != model.sample_rate:
        raise ValueError(f"audio must be resampled to {model.sample_rate} Hz first (got {sample_rate})")
    audio = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).float().reshape(-1)
    dev = next(model.parameters()).device
    x = torch.clamp(audio.to(dev).reshape(1, 1, -1), -1.0, 1.0)
    z, mu, logs = model.encode(x, sample_rate)
    z, mu, logs = z.squeeze(0).cpu(), mu.squeeze(0).cpu(), logs.squeeze(0).cpu()
    n = audio.numel()
    return {
        "z": z, "mu": mu, "logs": logs,
        "sample_rate": sample_rate,
        "compression_ratio": x.shape[-1] // z.shape[-1],
        "original_duration": n / sample_rate,
        "original_samples": n,
        "latent_shape": list(z.shape),
        "original_path": original_path,
    }


def save_latent(model, audio, sample_rate: int, audio_path: str) -> str:
    """Encodes and writes `<audio stem>_latent2x.pt` next to the audio file; returns the path written."""
    rec = latent_record(model, audio, sample_rate, audio_path)
    out = latent_path(audio_path)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    torch.save(rec, out)
    return out


def load_speech_latent(path: str, speech_token: List[int] = None, token_latent_ratio: int = 2) -> Tuple[torch.Tensor, List[int]]:
    """Reads a latent file as the TTS side does (processor.py:149-159): returns (speech_latent [T, D], speech_token),
    trimmed so that T == token_latent_ratio * len(speech_token) when a ratio and tokens are given."""
    rec = torch.load(path, map_location="cpu", weights_only=False)
    lat = rec["z"].transpose(0, 1)
    if token_latent_ratio != 0 and speech_token is not None:
        token_len = int(min(lat.shape[0] / token_latent_ratio, len(speech_token)))
        lat = lat[:token_latent_ratio * token_len]
        speech_token = speech_token[:token_len]
    return lat, speech_token
