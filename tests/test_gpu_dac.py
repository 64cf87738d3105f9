"""DAC-VAE decoder and encoder on the HIP path vs the golden vectors produced by the reference's own DACVAE
(tests/golden/dac*.npz, dacenc.npz) and vs the CPU oracle at the full BASELINE size."""
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


@pytest.mark.parametrize("dt,tol", [(1, 4e-2), (2, 2e-4)])
def test_dac_fused_residual_units_vs_two_launch_path(golden_dir, dt, tol):
    """mmx_dac_ru (one kernel per ResidualUnit of the 192- / 96- / 48-channel stages) against the same build running every
    unit as two windowed-GEMM launches: batch 2, 37 frames (ragged last tiles in every stage), and a 3-frame input (every
    stage shorter than one tile's halo).  Split build: the two paths differ by summation order only; bf16 build: by the
    rounding points of the intermediate activations (the fused kernel keeps the residual stream in fp32 throughout)."""
    from mmx.dac import DacDecoderEngine
    from oracle import weights as W
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_dac80.json")), SEED)
    fused, plain = DacDecoderEngine(sd, RATES, dtype=dt), DacDecoderEngine(sd, RATES, dtype=dt, fuse_ru=False)
    assert any(b["fused"] for b in fused.blocks) and not any(b["fused"] for b in plain.blocks)
    for B, T in ((2, 37), (1, 3)):
        z = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(T)).cuda()
        a, b = fused.decode(z), plain.decode(z)
        assert a.shape == b.shape == (B, 1, T * 480)
        err = (a - b).abs().max().item()
        assert err < tol, (dt, B, T, err)


@pytest.mark.parametrize("dt", [1, 2])
def test_dac_fused_unit_lengths(golden_dir, dt):
    """mmx_dac_ru with per-utterance lengths (a zero-padded batch): a member's valid rows equal the unit run on that member
    alone (rows beyond its length read as conv padding), its rows beyond the length are written as zero - x_out and act_out."""
    from mmx import ops
    from mmx._lib import TORCH_DT
    from mmx.dac import DacDecoderEngine
    from oracle import weights as W
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_dac80.json")), SEED)
    eng = DacDecoderEngine(sd, RATES, dtype=dt)
    g = torch.Generator().manual_seed(3)
    for blk in eng.blocks:
        if not blk["fused"]:
            continue
        C_, T, lens = blk["cout"], 300, [300, 173]
        x = torch.randn(2, T, C_, generator=g).cuda()
        lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
        for ru in blk["rus"]:
            xo, ao = torch.full_like(x, 7.0), torch.full((2, T, C_), 7.0, dtype=TORCH_DT[dt], device="cuda")
            ops.dac_ru(x, xo, ru, B=2, T=T, C_=C_, dil=ru["dil"], dtype=dt, act_out=ao, alpha_next=ru["a0"], lens=lens_t)
            for b, n in enumerate(lens):
                x1 = x[b:b + 1, :n].contiguous()
                xo1, ao1 = torch.empty_like(x1), torch.empty(1, n, C_, dtype=TORCH_DT[dt], device="cuda")
                ops.dac_ru(x1, xo1, ru, B=1, T=n, C_=C_, dil=ru["dil"], dtype=dt, act_out=ao1, alpha_next=ru["a0"])
                assert torch.equal(xo[b, :n], xo1[0]) and torch.equal(ao[b, :n], ao1[0]), (C_, ru["dil"], b)
                assert float(xo[b, n:].abs().max() if n < T else 0.0) == 0.0 and float(ao[b, n:].float().abs().max() if n < T else 0.0) == 0.0


# ----------------------------------------------------------------------------- encoder
ENC_RATES = [2, 3, 4, 4, 5]


def _enc_engine(golden_dir, dt):
    from mmx.dac import DacEncoderEngine
    from oracle import weights as W
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_dacenc.json")), SEED)
    return DacEncoderEngine(sd, ENC_RATES, dtype=dt), sd


@pytest.mark.parametrize("dt,tol", [(0, 2e-4), (1, 8e-2)])
def test_dac_encode_vs_reference_golden(golden_dir, dt, tol):
    """DACVAE.encode (dac-vae/model.py:469-483): mu / logs / z against the reference's outputs, with the reference's
    own randn draw as the noise.  fp32 build 2e-4 abs; bf16 build: operand rounding through 32 convs (bound here)."""
    from oracle import dac as ODAC
    eng, _ = _enc_engine(golden_dir, dt)
    g = np.load(os.path.join(golden_dir, "dacenc.npz"))
    for n in (4800, 11000):
        x = ODAC.preprocess(torch.from_numpy(g[f"wav_{n}"]), 480).cuda()
        z, mu, logs = eng.encode(x, noise=torch.from_numpy(g[f"noise_{n}"]).cuda())
        for name, v in (("mu", mu), ("logs", logs)):
            ref = torch.from_numpy(g[f"{name}_{n}"])
            assert v.shape == ref.shape
            err = (v.cpu() - ref).abs().max().item()
            assert err < tol, (name, n, dt, err)
        zr = torch.from_numpy(g[f"z_{n}"])
        rel = ((z.cpu() - zr).abs() / (1 + zr.abs())).max().item()       # z = mu + noise*exp(logs), |z| up to ~1e2
        assert rel < (2e-4 if dt == 0 else 0.3), (n, dt, rel)


def test_dac_encode_full_size_vs_oracle(golden_dir):
    """10 s of audio (240000 samples -> 500 frames), batch 2, fp32 build vs the CPU oracle; ragged second row is
    right-padded the way DACVAE.preprocess does."""
    from oracle import dac as ODAC
    eng, sd = _enc_engine(golden_dir, 0)
    g = torch.Generator().manual_seed(5)
    wav = 0.3 * torch.randn(2, 1, 240000, generator=g)
    wav[1, :, 200000:] = 0
    noise = torch.randn(2, 80, 500, generator=g)
    z, mu, logs = eng.encode(wav.cuda(), noise=noise.cuda())
    assert z.shape == (2, 80, 500)
    zr, mr, lr = ODAC.encode(sd, wav, ENC_RATES, noise)
    assert (mu.cpu() - mr).abs().max().item() < 3e-4
    assert (logs.cpu() - lr).abs().max().item() < 3e-4
    assert ((z.cpu() - zr).abs() / (1 + zr.abs())).max().item() < 3e-4


def test_dac_encode_unpadded_audio_vs_reference_golden(golden_dir):
    """dac-vae/extract_dac_latents.py:20-36 encodes clamped, UNPADDED audio: every strided conv floors its length."""
    eng, _ = _enc_engine(golden_dir, 0)
    g = np.load(os.path.join(golden_dir, "dacenc.npz"))
    x = torch.from_numpy(g["wav_11000"]).clamp(-1.0, 1.0).cuda()
    assert eng.frames(11000) == g["mu_raw_11000"].shape[-1]
    z, mu, logs = eng.encode(x, noise=torch.from_numpy(g["noise_raw_11000"]).cuda())
    assert (mu.cpu() - torch.from_numpy(g["mu_raw_11000"])).abs().max().item() < 2e-4
    assert (logs.cpu() - torch.from_numpy(g["logs_raw_11000"])).abs().max().item() < 2e-4
    zr = torch.from_numpy(g["z_raw_11000"])
    assert ((z.cpu() - zr).abs() / (1 + zr.abs())).max().item() < 2e-4
    with pytest.raises(ValueError, match="shorter than one latent frame"):
        eng.encode(torch.zeros(1, 1, 100, device="cuda"))


def test_strided_windowed_gemm_matches_conv1d():
    """MmxGemmParams.row_stride: Conv1d(k=2s, stride s, pad ceil(s/2)) for the encoder strides, fp32 build, vs torch."""
    import math
    from mmx import ops
    g = torch.Generator().manual_seed(9)
    for s, cin, cout, T in ((2, 64, 128, 960), (3, 16, 40, 33 * 3), (5, 24, 48, 5 * 41)):
        w = torch.randn(cout, cin, 2 * s, generator=g) / math.sqrt(cin * 2 * s)
        b = torch.randn(cout, generator=g)
        x = torch.randn(2, cin, T, generator=g)
        ref = torch.nn.functional.conv1d(x, w, b, stride=s, padding=math.ceil(s / 2))
        out = torch.empty(2, T // s, cout, device="cuda")
        ops.conv1d(x.transpose(1, 2).contiguous().cuda(), ops.pack_conv1d(w.cuda(), 0), T=T, Cin=cin, k=2 * s,
                   pad_left=math.ceil(s / 2), stride=s, T_out=T // s, dtype=0, batch=2, bias=b.cuda(), out_f32=out)
        assert ref.shape[-1] == T // s
        assert (out.cpu().transpose(1, 2) - ref).abs().max().item() < 1e-4, s


@pytest.mark.parametrize("dt", [0, 1, 2])
def test_dac_decode_of_a_zero_padded_batch_equals_each_member(golden_dir, dt):
    """DacDecoderEngine.decode_time_major(lens=...): ONE decode of a zero-padded batch of latents of different lengths (what a flow
    group hands over) gives every member the samples of its own decode, bit for bit - the row masks of the GEMM epilogues, the mask
    launch behind each unfused ConvTranspose1d (mmx_mask_rows) and the lengths of the fused ResidualUnits stand in for the zero
    padding a member decoded alone would see (dac-vae/model.py:107-143,252-284).  Lengths chosen so that members end inside tiles."""
    from mmx import ops
    from mmx._lib import TORCH_DT
    eng, _ = _engine(golden_dir, 80, dt)
    lens = [37, 9, 50, 23]
    g = torch.Generator().manual_seed(3)
    lat = [torch.randn(n, 80, generator=g).cuda().to(TORCH_DT[dt]) for n in lens]
    Tm = max(lens)
    zt = torch.zeros(len(lens), Tm, 80, dtype=TORCH_DT[dt], device="cuda")
    for i, l in enumerate(lat):
        zt[i, :lens[i]] = l
    wav = eng.decode_time_major(zt, len(lens), Tm, lens=lens)
    assert wav.shape == (len(lens), 1, Tm * 480)
    for i, l in enumerate(lat):
        one = eng.decode_time_major(l.reshape(1, lens[i], 80).contiguous(), 1, lens[i])
        assert torch.equal(wav[i:i + 1, :, :lens[i] * 480], one), (dt, i, (wav[i:i + 1, :, :lens[i] * 480] - one).abs().max().item())
