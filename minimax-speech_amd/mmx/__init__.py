"""mmx — Python host side of the MI355X-native TTS hot path (Qwen2 AR LM -> CosyVoice2 flow -> DAC-VAE).

Everything that computes goes through libmmx_hip.so (hand-written HIP for gfx950, C ABI in
include/mmx_hip.h).  PyTorch is used for device memory, streams, graphs and torch.distributed only.
There is NO CPU fallback: importing works anywhere (so CPU-only tests can check the ABI), but every op
raises if the library or a GPU is missing.
"""
from . import _lib  # noqa: F401
