"""DAC-VAE decoder on the HIP path vs the golden vectors produced by the reference's own Decoder
(tests/golden/dac*.npz) and vs the CPU oracle at the full BASELINE size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RATES = [5, 4, 4, 3, 2]
SEED = 7


def _engine(golden_dir, lat, dt):
    from mmx.dac import DacDecoderEngine
    from oracle import weights as W
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, f"manifest_dac{lat}.json")), SEED)
    return DacDecoderEngine(sd, RATES, dtype=dt), sd


@pytest.mark.parametrize("lat", [80, 128])
@pytest.mark.parametrize("dt,tol", [(0, 1e-4), (1, 3e-2)])
def test_dac_decode_vs_reference_golden(golden_dir, lat, dt, tol):
    """BASELINE config 1 (dac-vae Decoder, 128-d and 80-d latents). fp32 build: within 1e-4 abs of the reference
    (north-star bound 1e-3); bf16 build: bf16 operand rounding through 37 convs, bound stated here."""
    eng, _ = _engine(golden_dir, lat, dt)
    g = np.load(os.path.join(golden_dir, f"dac{lat}.npz"))
    for T in (8, 50):
        z = torch.from_numpy(g[f"z_T{T}"]).cuda()
        wav = eng.decode(z)
        ref = torch.from_numpy(g[f"wav_T{T}"])
        assert wav.shape == ref.shape
        err = (wav.cpu() - ref).abs().max().item()
        assert err < tol, (lat, dt, T, err)


def test_dac_decode_full_size_vs_oracle(golden_dir):
    """10 s utterance (500 frames -> 240000 samples), batch 2, fp32 build vs the CPU oracle."""
    from oracle import dac as ODAC
    eng, sd = _engine(golden_dir, 80, 0)
    z = torch.randn(2, 80, 500, generator=torch.Generator().manual_seed(1))
    wav = eng.decode(z.cuda())
    assert wav.shape == (2, 1, 240000)
    ref = ODAC.decode(sd, z, RATES)
    assert (wav.cpu() - ref).abs().max().item() < 2e-4
