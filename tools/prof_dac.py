"""Three DAC-VAE decodes (500 latent frames -> 10 s of audio, bf16) for kernel-trace / PMC profiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import shapes, synth
from mmx.dac import DacDecoderEngine
dec = DacDecoderEngine(synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0), [5, 4, 4, 3, 2])
z = torch.randn(1, 80, 500, device="cuda")
for _ in range(3):
    dec.decode(z)
torch.cuda.synchronize()
