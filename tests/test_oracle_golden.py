"""Pins the CPU oracle (oracle/*.py) to the golden vectors produced by the REFERENCE's own classes
(oracle/gen_golden.py, run in the build container; fixtures under tests/golden/).
The reference holds no tests or known-answer vectors for this path."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dac as ODAC
from oracle import flow as OFLOW
from oracle import llm as OLLM
from oracle import weights as W

SEED = 7
RATES = [5, 4, 4, 3, 2]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("lat", [80, 128])
def test_dac_decode_matches_reference(golden_dir, lat):
    g = _load(golden_dir, f"dac{lat}.npz")
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, f"manifest_dac{lat}.json")), SEED)
    for T in (8, 50):
        y = ODAC.decode(sd, torch.from_numpy(g[f"z_T{T}"]), RATES)
        ref = torch.from_numpy(g[f"wav_T{T}"])
        assert y.shape == ref.shape == (1, 1, T * 480)
        assert (y - ref).abs().max() < 2e-5, float((y - ref).abs().max())
    # stage activations
    x = ODAC.wnconv1d_act(sd, "de_conv_pre", torch.from_numpy(g["z_T8"]))
    assert torch.allclose(x, torch.from_numpy(g["pre_T8"]), atol=1e-5)
    _, st = ODAC.decoder_forward(sd, x, RATES, return_stages=True)
    for i in range(3):
        assert torch.allclose(st[i], torch.from_numpy(g[f"stage{i}_T8"]), atol=2e-5), i


def test_dac_decode_trained_weight_norm_matches_reference(golden_dir):
    """The "fp32" checkpoint kind (mmx/synth.py): weight_g = s * ||weight_v|| with per-channel s in [0.6, 1.5], folded weights
    that are general fp32 values — the reference's Decoder on it (dac80_fp32.npz).  A fold that ignored weight_g, or took the
    ConvTranspose1d's norm over the wrong axis (its weight_g is [Cin, 1, 1], dac-vae/layers.py:13-14), fails here: both are
    tried below and must miss the golden by far more than the tolerance."""
    g = _load(golden_dir, "dac80_fp32.npz")
    man = W.load_manifest(os.path.join(golden_dir, "manifest_dac80.json"))
    sd = W.synth_state_dict(man, SEED, kind="fp32")
    k0 = "decoder.model.1.block.1."
    ratio = sd[k0 + "weight_g"].flatten() / torch.norm_except_dim(sd[k0 + "weight_v"], 2, 0).flatten()
    assert sd[k0 + "weight_g"].shape == (sd[k0 + "weight_v"].shape[0], 1, 1) and ratio.min() < 0.7 and ratio.max() > 1.4
    for T in (8, 50):
        y = ODAC.decode(sd, torch.from_numpy(g[f"z_T{T}"]), RATES)
        ref = torch.from_numpy(g[f"wav_T{T}"])
        assert y.shape == ref.shape and (y - ref).abs().max() < 2e-5, float((y - ref).abs().max())
    x = ODAC.wnconv1d_act(sd, "de_conv_pre", torch.from_numpy(g["z_T8"]))
    _, st = ODAC.decoder_forward(sd, x, RATES, return_stages=True)
    for i in range(3):
        assert torch.allclose(st[i], torch.from_numpy(g[f"stage{i}_T8"]), atol=2e-5), i
    ref = torch.from_numpy(g["wav_T8"])
    # (a) weight_g ignored
    bad = dict(sd)
    for k in man:
        if k.endswith(".weight_g"):
            bad[k] = torch.norm_except_dim(sd[k[:-1] + "v"], 2, 0).reshape(sd[k].shape)
    assert (ODAC.decode(bad, torch.from_numpy(g["z_T8"]), RATES) - ref).abs().max() > 1e-2
    # (b) the ConvTranspose1d's gain applied per OUTPUT channel (axis 1) instead of per dim-0 slice
    bad = dict(sd)
    v, gg = sd[k0 + "weight_v"], sd[k0 + "weight_g"]
    wrong = v * (gg.flatten()[:v.shape[1]].reshape(1, -1, 1) / v.norm(dim=(0, 2), keepdim=True))
    bad[k0 + "weight_v"], bad[k0 + "weight_g"] = wrong, torch.norm_except_dim(wrong, 2, 0)
    assert (ODAC.decode(bad, torch.from_numpy(g["z_T8"]), RATES) - ref).abs().max() > 1e-2


def test_dac_encode_matches_reference(golden_dir):
    """DACVAE.encode / forward (dac-vae/model.py:469-506) on the reference's own outputs."""
    g = _load(golden_dir, "dacenc.npz")
    man = W.load_manifest(os.path.join(golden_dir, "manifest_dacenc.json"))
    man.update(W.load_manifest(os.path.join(golden_dir, "manifest_dac80.json")))
    sd = W.synth_state_dict(man, SEED)
    enc_rates = [2, 3, 4, 4, 5]
    for n in (4800, 11000):
        wav = torch.from_numpy(g[f"wav_{n}"])
        x = ODAC.preprocess(wav, 480)
        assert x.shape[-1] % 480 == 0
        z, mu, logs = ODAC.encode(sd, x, enc_rates, torch.from_numpy(g[f"noise_{n}"]))
        for name, v in (("mu", mu), ("logs", logs)):
            assert (v - torch.from_numpy(g[f"{name}_{n}"])).abs().max() < 2e-5, name
        zr = torch.from_numpy(g[f"z_{n}"])
        assert ((z - zr).abs() / (1 + zr.abs())).max() < 2e-5
        recon = ODAC.decode(sd, z, RATES)[..., :n]
        assert (recon - torch.from_numpy(g[f"recon_{n}"])).abs().max() < 5e-5
    zr, mur, _ = ODAC.encode(sd, torch.from_numpy(g["wav_11000"]).clamp(-1, 1), enc_rates, torch.from_numpy(g["noise_raw_11000"]))
    assert mur.shape == g["mu_raw_11000"].shape and (mur - torch.from_numpy(g["mu_raw_11000"])).abs().max() < 2e-5
    _, st = ODAC.encoder_forward(sd, ODAC.preprocess(torch.from_numpy(g["wav_4800"]), 480), enc_rates, return_stages=True)
    for i in (0, 1, 5):
        assert torch.allclose(st[i][..., :160], torch.from_numpy(g[f"stage{i}_4800"]), atol=2e-5), i


@pytest.fixture(scope="module")
def flow_sd(golden_dir):
    return W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_flow.json")), SEED)


def test_flow_rand_noise(golden_dir):
    g = _load(golden_dir, "flow.npz")
    assert np.array_equal(OFLOW.rand_noise()[:, :, :64].numpy(), g["rand_noise_head"])


def test_flow_estimator_matches_reference(golden_dir, flow_sd):
    g = _load(golden_dir, "flow.npz")
    t = lambda k: torch.from_numpy(g[k])
    T = 64
    mask = torch.ones(2, 1, T)
    for name, streaming, mk in (("est_full", False, mask), ("est_stream", True, mask),
                                ("est_padmask", False, t("est_mask2"))):
        y = OFLOW.estimator_forward(flow_sd, "decoder.estimator", t("est_x"), mk, t("est_mu"), t("est_t"),
                                    t("est_spks"), t("est_cond"), streaming)
        err = (y - t(name)).abs().max()
        assert err < 2e-5, (name, float(err))


def test_flow_encoder_matches_reference(golden_dir, flow_sd):
    g = _load(golden_dir, "flow.npz")
    xs, ctx = torch.from_numpy(g["enc_xs"]), torch.from_numpy(g["enc_ctx"])
    h, _ = OFLOW.encoder_forward(flow_sd, "encoder", xs, torch.tensor([25]), None, False)
    assert (h - torch.from_numpy(g["enc_full"])).abs().max() < 2e-5
    h, _ = OFLOW.encoder_forward(flow_sd, "encoder", xs, torch.tensor([25]), ctx, True)
    assert (h - torch.from_numpy(g["enc_ctx_stream"])).abs().max() < 2e-5


def test_flow_inference_matches_reference(golden_dir, flow_sd):
    g = _load(golden_dir, "flow.npz")
    tok, ptok = torch.from_numpy(g["flow_tok"]), torch.from_numpy(g["flow_ptok"])
    pfeat, emb = torch.from_numpy(g["flow_pfeat"]), torch.from_numpy(g["flow_emb"])
    none_tok, none_feat = torch.zeros(1, 0, dtype=torch.long), torch.zeros(1, 0, 80)
    cases = {
        "flow_noprompt": (tok, none_tok, none_feat, False, True),
        "flow_prompt": (tok, ptok, pfeat, False, True),
        "flow_stream_nofinal": (tok, ptok, pfeat, True, False),
        "flow_stream_final": (tok, ptok, pfeat, True, True),
    }
    for name, (tk, pt, pf, streaming, finalize) in cases.items():
        y = OFLOW.flow_inference(flow_sd, tk, pt, pf, emb, streaming, finalize)
        ref = torch.from_numpy(g[name])
        assert y.shape == ref.shape
        assert (y - ref).abs().max() < 1e-4, (name, float((y - ref).abs().max()))


def test_llm_teacher_forced_logp_matches_reference(golden_dir):
    """Full-size (24-layer, 494 M parameter) Qwen2 backbone: prefill + 16 teacher-forced decode steps."""
    g = _load(golden_dir, "llm.npz")
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_llm.json")), SEED)
    cfg = OLLM.QwenCfg()
    lm_input = OLLM.build_lm_input(sd, torch.from_numpy(g["text"]), torch.from_numpy(g["ptext"]),
                                   torch.from_numpy(g["pspeech"]))
    assert torch.allclose(lm_input, torch.from_numpy(g["lm_input"]), atol=1e-6)
    x, cache = lm_input, None
    for i in range(17):
        y, cache = OLLM.qwen2_forward(sd, cfg, x, cache)
        if i == 0:
            assert (y - torch.from_numpy(g["prefill_hidden"])).abs().max() < 1e-4
        logp = torch.nn.functional.linear(y[:, -1], sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(-1)[0]
        err = (logp - torch.from_numpy(g["logp"][i])).abs().max()
        assert err < 2e-4, (i, float(err))
        if i < 16:
            x = sd["speech_embedding.weight"][int(g["forced"][i])].reshape(1, 1, -1)


def test_sampler_matches_reference_under_torch_seeds(golden_dir):
    """ras/nucleus/random sampling with torch's own generator as the noise source: ids identical to the
    reference's (common.py:111-139) for 200 recorded seeds."""
    g = _load(golden_dir, "sampler.npz")
    for s, (logp, hist) in enumerate(W.sampler_cases()):
        torch.manual_seed(1000 + s)
        assert OLLM.ras_sampling_e(logp, hist, OLLM.torch_noise) == int(g["ras"][s]), s
        torch.manual_seed(1000 + s)
        p, idx = OLLM.nucleus_candidates(logp)
        assert int(idx[OLLM.multinomial_e(p, OLLM.torch_noise(0, p.numel()))]) == int(g["nucleus"][s]), s
        torch.manual_seed(1000 + s)
        pr = logp.softmax(0)
        assert OLLM.multinomial_e(pr, OLLM.torch_noise(1, 6564)) == int(g["random"][s]), s


def test_multinomial_is_exponential_race():
    """torch.multinomial(p, 1) == argmax(p / Exp(1) noise) on the same generator stream."""
    for s in range(200):
        n = 25 if s % 2 else 6564
        p = torch.rand(n, generator=torch.Generator().manual_seed(s)) ** 3
        torch.manual_seed(s)
        a = int(p.multinomial(1, replacement=True))
        torch.manual_seed(s)
        assert OLLM.multinomial_e(p, OLLM.torch_noise(0, n)) == a


def test_speaker_encoder_matches_reference(golden_dir):
    """LearnableSpeakerEncoder and flow.inference with reference mels (use_speaker_encoder=True)."""
    from oracle import spk as OSPK
    g = _load(golden_dir, "spk.npz")
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_spk.json")), SEED)
    for T in (37, 150):
        e = OSPK.speaker_encoder(sd, torch.from_numpy(g[f"mel_T{T}"]))
        assert (e - torch.from_numpy(g[f"emb_T{T}"])).abs().max() < 2e-5
    fsd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_flow_spk.json")), SEED)
    emb = OSPK.reference_embedding(fsd, torch.from_numpy(g["flow_refs"]))
    y = OFLOW.flow_inference(fsd, torch.from_numpy(g["flow_tok"]), torch.zeros(1, 0, dtype=torch.long), torch.zeros(1, 0, 80), emb)
    assert (y - torch.from_numpy(g["flow_out"])).abs().max() < 1e-4


def _scripted_ids(samp, eos=6561):
    """sampling_ids (llm.py:259-274) around a `sampling` callable."""
    def f(logp, out, call, ignore_eos):
        for _ in range(101):
            top = int(samp(logp, out, 25))
            if (not ignore_eos) or top != eos:
                return top
        raise RuntimeError("max_trials")
    return f


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_llm_bistream_matches_reference(golden_dir, case):
    """Qwen2LM.inference_bistream (llm.py:762-870) run by the reference with a scripted sampler: yielded tokens and
    the log-probs seen by every sampling call."""
    from mmx import shapes
    g = _load(golden_dir, "bistream.npz")
    sd = W.synth_state_dict(shapes.llm_manifest(layers=2), SEED)
    cfg = OLLM.QwenCfg(layers=2)
    rec = []
    samp = OLLM.ScriptedSampling(tuple(g[f"{case}_fill_at"].tolist()), int(g[f"{case}_eos_from"]), record=rec)
    text = torch.from_numpy(g[f"{case}_text"])
    chunks, o = [], 0
    for n in g[f"{case}_chunks"].tolist():
        chunks.append(text[:, o:o + n])
        o += n
    toks, hist = OLLM.lm_inference_bistream(sd, cfg, chunks, torch.from_numpy(g[f"{case}_ptext"]),
                                            torch.from_numpy(g[f"{case}_pspeech"]), _scripted_ids(samp))
    assert toks == g[f"{case}_tokens"].tolist()
    assert hist[-1] == 6561 and [t for t in hist if t < 6561] == toks
    lp = torch.stack(rec)
    assert lp.shape[0] == g[f"{case}_logp_head"].shape[0]
    assert (lp[:, :128] - torch.from_numpy(g[f"{case}_logp_head"])).abs().max() < 2e-4
    assert (lp.max(dim=1).values - torch.from_numpy(g[f"{case}_logp_max"])).abs().max() < 2e-4


def test_flow_sub_blocks_match_reference(golden_dir, flow_sd):
    """Every estimator sub-block of the oracle vs the reference's own modules (tests/golden/blocks.npz): time MLP, causal
    block, causal ResNet block (256- and 320-channel input), transformer block under a pad mask and under a chunk mask."""
    g = _load(golden_dir, "blocks.npz")
    t = lambda k: torch.from_numpy(g[k])
    e = "decoder.estimator"
    temb = OFLOW.sinusoidal_pos_emb(torch.tensor([0.25, 0.7]), 320)
    assert (temb - t("temb_in")).abs().max() < 1e-5
    te = F.linear(F.silu(F.linear(temb, flow_sd[e + ".time_mlp.linear_1.weight"], flow_sd[e + ".time_mlp.linear_1.bias"])),
                  flow_sd[e + ".time_mlp.linear_2.weight"], flow_sd[e + ".time_mlp.linear_2.bias"])
    assert (te - t("temb")).abs().max() < 2e-5
    x, m = t("cb_x").transpose(1, 2), t("cb_mask").transpose(1, 2)
    assert (OFLOW.causal_block(flow_sd, e + ".final_block", x, m).transpose(1, 2) - t("cb_out")).abs().max() < 2e-5
    assert (OFLOW.causal_resnet(flow_sd, e + ".mid_blocks.0.0", x, m, t("temb")).transpose(1, 2) - t("rb_out")).abs().max() < 2e-5
    y = OFLOW.causal_resnet(flow_sd, e + ".down_blocks.0.0", t("rb0_x").transpose(1, 2), m, t("temb")).transpose(1, 2)
    assert (y - t("rb0_out")).abs().max() < 2e-5
    for name in ("pad", "chunk"):
        y = OFLOW.basic_transformer_block(flow_sd, e + ".mid_blocks.0.1.0", t("tb_hs"), t(f"tb_bias_{name}"))
        assert (y - t(f"tb_out_{name}")).abs().max() < 2e-5, name



def test_stream_schedule_restates_reference_hops():
    """oracle/stream.py hop_schedule against cli/model.py:336-378 worked by hand: (visible tokens, token_offset, finalize)."""
    from oracle import stream as OS
    # no prompt, 118 tokens: hops of 25 once 28 tokens past the offset exist, then the closing pass over everything
    assert OS.hop_schedule(118, 0) == [(28, 0, False), (53, 25, False), (78, 50, False), (103, 75, False), (118, 100, True)]
    # 30 prompt tokens: prompt_token_pad = ceil(30 / 25) * 25 - 30 = 20 -> the first hop takes 45 tokens
    assert OS.hop_schedule(130, 30) == [(48, 0, False), (73, 45, False), (98, 70, False), (123, 95, False), (130, 120, True)]
    # too short for a single hop: only the closing pass
    assert OS.hop_schedule(20, 0) == [(20, 0, True)]
    # exactly hop + look-ahead tokens: one hop, then the closing pass over the same tokens
    assert OS.hop_schedule(28, 0) == [(28, 0, False), (28, 25, True)]


def test_stream_schedule_matches_reference_loop(golden_dir):
    """hop_schedule against the calls the REFERENCE's own CosyVoice2Model.tts(stream=True) loop made when driven with stub
    llm / flow / hift objects (oracle/gen_golden.py::gen_stream, tests/golden/stream.npz): per flow call (tokens seen,
    token_offset as recovered from the mel handed to the vocoder, finalize, streaming flag); the closing pass runs with
    streaming=False (cli/model.py:371-378 leaves `stream` at its default)."""
    from oracle import stream as OS
    g = _load(golden_dir, "stream.npz")
    assert len(g["cases"]) >= 5
    for N, Lp in g["cases"].tolist():
        ref = g[f"calls_{N}_{Lp}"].tolist()
        got = OS.hop_schedule(N, Lp)
        assert [(v, o, bool(f)) for v, o, f, _ in ref] == got, (N, Lp)
        assert all(st == (not f) for _, _, f, st in ref), (N, Lp)        # streaming flag of every call: not finalize


def test_stream_rendering_rule_is_exact_when_passes_agree(golden_dir):
    """The DAC context rule of oracle/stream.py (restated from mmx/pipeline.py::tts_stream): when every pass holds the same
    latents (streaming passes do; here the closing pass too), hold-back, context windows and the unit-sum cross-fade must
    reproduce ONE offline decode of the latents sample for sample."""
    from mmx.dac import DacDecoderEngine
    from oracle import dac as ODAC, stream as OS
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_dac80.json")), SEED)
    rates = [5, 4, 4, 3, 2]
    CL, CR = DacDecoderEngine.receptive_field(rates)
    lat = torch.randn(130, 80, generator=torch.Generator().manual_seed(3))
    passes = [(0, lat[:50], False), (50, lat[:100], False), (100, lat, True)]
    with torch.no_grad():
        got = torch.cat(OS.render_passes(sd, rates, passes, CL, CR))
        ref = ODAC.decode(sd, lat.t().unsqueeze(0).contiguous(), rates)[0, 0]
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-6
    w = OS.fade_window(8 * 480)
    assert (w[:3840] + w[3840:] - 1).abs().max().item() < 1e-6
