"""Times the DAC-VAE encoder and decoder engines (bf16 build) on synthetic weights: ms per call and audio-s/s."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
from mmx import shapes, synth  # noqa: E402
from mmx.dac import DacDecoderEngine, DacEncoderEngine  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    enc = DacEncoderEngine(synth.synth_state_dict(shapes.dac_encoder_manifest(80), 0), [2, 3, 4, 4, 5])
    dec = DacDecoderEngine(synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0), [5, 4, 4, 3, 2])
    for B in (1, 8):
        wav = 0.3 * torch.randn(B, 1, 240000, device="cuda")
        z = torch.randn(B, 80, 500, device="cuda")
        te = timeit(lambda: enc.encode(wav))
        td = timeit(lambda: dec.decode(z))
        print(f"B={B} 10 s each: encode {te:.2f} ms ({B * 10 / te * 1e3:.0f} audio-s/s)  decode {td:.2f} ms "
              f"({B * 10 / td * 1e3:.0f} audio-s/s)", flush=True)


if __name__ == "__main__":
    main()
