"""Composed hot path at BASELINE config-3 size — LM (24 layers, 250 forced steps) -> flow (500 frames x 10 Euler
steps, CFG) -> DAC-VAE decoder (240 000 samples) — on the HIP path vs the CPU oracle's composed path
(speech/cosyvoice/cli/model.py:285-319,321-386 with hift.inference replaced by DACVAE.decode; llm.py:745-760).

North star: FSQ token ids bit-exact and waveform within 1e-3 abs on identical inputs.  The fp32 build is held to exactly
that.  The bf16 build cannot be (bf16 rounding of GEMM inputs through 24 LM layers / 70 estimator blocks x 10 steps on
random-init weights): its bound is stated as a requirement relative to the signal — id agreement with the oracle over the
free-running decode, waveform SNR given the oracle's token ids — and the measured numbers are printed for DESIGN.md.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

N_TEXT, N_STEPS, SEED = 48, 250, 0
WAV_TOL_F32 = 1e-3                 # north star (abs, waveform in [-1, 1])
BF16_MIN_SNR_DB = 25.0             # bf16 build, same token ids: waveform SNR vs the oracle (requirement, not a measurement)
BF16_MIN_TF_AGREEMENT = 0.6        # bf16 build, teacher forced with the oracle's ids: share of steps that draw the same id


@pytest.fixture(scope="module")
def case():
    """Config-3 inputs and the oracle's composed outputs (about 20 s of CPU work)."""
    from mmx import shapes, synth
    from oracle import dac as ODAC, flow as OFLOW, llm as OLLM
    llm_sd = synth.synth_state_dict(shapes.llm_manifest(), 0)
    flow_sd = synth.synth_state_dict(shapes.flow_manifest(), 0)
    dac_sd = synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0)
    text = torch.randint(0, 151936, (1, N_TEXT), generator=torch.Generator().manual_seed(2))
    emb = torch.randn(1, 192, generator=torch.Generator().manual_seed(1))
    z = torch.zeros(1, 0, dtype=torch.long)
    with torch.no_grad():
        toks = OLLM.lm_inference(llm_sd, OLLM.QwenCfg(), text, z, z, seed=SEED, seq=0, max_steps=N_STEPS,
                                 ignore_eos_always=True)
        lat = OFLOW.flow_inference(flow_sd, torch.tensor(toks).reshape(1, -1), z, torch.zeros(1, 0, 80), emb)
        wav = ODAC.decode(dac_sd, lat, [5, 4, 4, 3, 2])
    return dict(llm_sd=llm_sd, flow_sd=flow_sd, dac_sd=dac_sd, text=text, emb=emb, toks=toks, lat=lat, wav=wav)


def _engine(case, dt):
    from mmx.pipeline import TtsEngine
    return TtsEngine(case["llm_sd"], case["flow_sd"], case["dac_sd"], dtype=dt, max_batch=1, max_ctx=640)


def _snr_db(ref, got):
    return float(10 * torch.log10(ref.pow(2).mean() / (ref - got).pow(2).mean().clamp_min(1e-30)))


def test_composed_pipeline_fp32_ids_identical_waveform_1e3(case):
    """fp32 build, free running: the same 250 ids as the CPU path, the waveform within 1e-3 abs."""
    eng = _engine(case, 0)
    for rep in range(2):                                   # eager pass, then the recorded graphs
        wav = eng.tts(case["text"].cuda(), case["emb"].cuda(), seed=SEED, exact_steps=N_STEPS)
        got = eng.llm.tokens()[0]
        assert got == case["toks"], ("token ids differ from the oracle", rep,
                                     next(i for i, (a, b) in enumerate(zip(got, case["toks"])) if a != b))
        assert wav.shape == case["wav"].shape == (1, 1, 2 * len(got) * 480)
        err = (wav.cpu() - case["wav"]).abs().max().item()
        print(f"fp32 composed: {len(got)} ids identical, waveform max abs err {err:.3e} "
              f"(std {case['wav'].std().item():.3f}, SNR {_snr_db(case['wav'], wav.cpu()):.1f} dB)")
        assert err <= WAV_TOL_F32, err


def test_composed_pipeline_bf16_bound(case):
    """bf16 build: (a) ids vs the oracle.  Free running, one flipped near-tie changes the whole continuation (random-init
    logits are flat), so the free-running mismatch count and first divergence are reported, and the requirement is put
    on the TEACHER-FORCED run (the oracle's ids are fed back, every step still draws its own id from the same Philox
    noise): the share of steps whose draw equals the oracle's.  (b) flow + DAC on the ORACLE's ids: waveform SNR / max
    abs error vs the oracle's waveform."""
    eng = _engine(case, 1)
    eng.tts(case["text"].cuda(), case["emb"].cuda(), seed=SEED, exact_steps=N_STEPS)
    got, want = eng.llm.tokens()[0], case["toks"]
    n = min(len(got), len(want))
    first = next((i for i in range(n) if got[i] != want[i]), n)
    mism = sum(1 for i in range(n) if got[i] != want[i]) + abs(len(got) - len(want))
    z0 = torch.zeros(1, 0, dtype=torch.long, device="cuda")
    x = eng.llm.build_lm_input(case["text"].cuda(), z0, z0)
    eng.llm.start([x], [N_STEPS], [N_STEPS], seed=SEED, forced=torch.tensor(want).reshape(1, -1))
    eng.llm.run(N_STEPS)
    drawn = eng.llm.sampled[0, :len(want)].tolist()
    agree = sum(1 for a, b in zip(drawn, want) if a == b) / len(want)
    tok = torch.tensor(want, device="cuda").reshape(1, -1)
    z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
    wav = eng.token2wav(tok, z, torch.zeros(1, 0, 80, device="cuda"), case["emb"].cuda()).cpu()
    err = (wav - case["wav"]).abs().max().item()
    snr = _snr_db(case["wav"], wav)
    print(f"bf16 composed: free running {mism}/{n} ids differ (first divergence at step {first}); teacher forced "
          f"{agree * 100:.1f} % of the draws equal the oracle's; same-ids waveform max abs err {err:.3e}, SNR {snr:.1f} dB "
          f"(std {case['wav'].std().item():.3f})")
    assert agree >= BF16_MIN_TF_AGREEMENT, agree
    assert wav.shape == case["wav"].shape and snr >= BF16_MIN_SNR_DB, (snr, err)


def test_drop_in_modules_compose_like_the_engine(case):
    """The reference-shaped modules (Qwen2LM / CausalMaskedDiffWithXvec / DACVAE under CosyVoice2Model.tts) give the same
    ids and waveform as the engine-level path in the fp32 build — the boundary adds no arithmetic."""
    from cosyvoice.cli.model import CosyVoice2Model
    from test_dropin_api import build_dac, build_flow, build_llm     # the config.yaml / configx2.yml instantiations
    llm, flow, dac = build_llm(24), build_flow(), build_dac(80)
    llm.load_state_dict(case["llm_sd"], strict=True)
    flow.load_state_dict(case["flow_sd"], strict=True)
    dac.load_state_dict(case["dac_sd"], strict=False)          # decoder half only (the encoder is not on this path)
    for m in (llm, flow, dac):
        m.to("cuda").eval()
        m.float_parity()
    llm.seed = SEED
    model = CosyVoice2Model(llm, flow, dac)
    # exactly N_STEPS decode steps with EOS ignored: min == max token/text ratio
    ratio = (N_STEPS + 0.5) / N_TEXT                        # int(48 * ratio) == 250 on both bounds
    orig = llm.inference
    llm.inference = lambda **kw: orig(**{**kw, "min_token_text_ratio": ratio, "max_token_text_ratio": ratio})
    out = list(model.tts(text=case["text"], flow_embedding=case["emb"], llm_embedding=case["emb"]))
    wav = torch.cat([o["tts_speech"] for o in out], dim=1)
    err = (wav - case["wav"][0]).abs().max().item()
    print(f"drop-in composed (fp32): waveform max abs err {err:.3e}")
    assert wav.shape[1] == case["wav"].shape[2] and err <= WAV_TOL_F32, err


def test_config4_rank_share_full_size(case):
    """BASELINE config 4, one rank's share at full size: 32 utterances, lengths U{50..500} tokens (seed 3), 24-layer LM.
    (a) bf16: the overlapped schedule (decode loop + flow / DAC workers on other streams, compaction to the 16-slot engine)
    produces the token ids of the back-to-back schedule, utterance by utterance;
    (b) fp32: the two shortest utterances out of the overlapped 32-batch equal the CPU oracle's composed path — ids
    identical, waveform within 1e-3 (sequence id = position in the batch keys the Philox stream on both sides)."""
    from mmx.pipeline import TtsEngine
    from oracle import dac as ODAC, flow as OFLOW, llm as OLLM
    lens = torch.randint(50, 501, (32,), generator=torch.Generator().manual_seed(3)).tolist()
    g = torch.Generator().manual_seed(2)
    texts = [torch.randint(0, 151936, (1, 48), generator=g) for _ in range(32)]
    emb = case["emb"].cuda()
    tc = [t.cuda() for t in texts]
    eng = TtsEngine(case["llm_sd"], case["flow_sd"], case["dac_sd"], dtype=1, max_batch=32, max_ctx=640)
    ref = [w.clone() for w in eng.tts_batch(tc, [emb] * 32, seed=0, exact_steps=lens, overlap=False)]
    want = [t.tolist() for t in eng.last_tokens]
    for rep in range(2):
        wavs = eng.tts_batch(tc, [emb] * 32, seed=0, exact_steps=lens, overlap=True)
        torch.cuda.synchronize()
        assert [t.tolist() for t in eng.last_tokens] == want, rep
        for b in range(32):                                  # same ids; other flow groups -> bf16 rounding only
            assert wavs[b].shape == ref[b].shape and wavs[b].shape[-1] == 2 * len(want[b]) * 480
            assert (wavs[b] - ref[b]).abs().max().item() < 3e-2, (rep, b)
    del eng
    torch.cuda.empty_cache()
    ef = TtsEngine(case["llm_sd"], case["flow_sd"], case["dac_sd"], dtype=0, max_batch=32, max_ctx=640)
    wf = ef.tts_batch(tc, [emb] * 32, seed=0, exact_steps=lens, overlap=True)
    torch.cuda.synchronize()
    z = torch.zeros(1, 0, dtype=torch.long)
    for b in sorted(range(32), key=lambda i: lens[i])[:2]:
        with torch.no_grad():
            toks = OLLM.lm_inference(case["llm_sd"], OLLM.QwenCfg(), texts[b], z, z, seed=0, seq=b, max_steps=lens[b], ignore_eos_always=True)
            lat = OFLOW.flow_inference(case["flow_sd"], torch.tensor(toks).reshape(1, -1), z, torch.zeros(1, 0, 80), case["emb"])
            wav = ODAC.decode(case["dac_sd"], lat, [5, 4, 4, 3, 2])
        err = (wf[b].cpu() - wav).abs().max().item()
        print(f"config-4 share, utterance {b} ({lens[b]} steps, {len(toks)} ids): fp32 waveform max abs err {err:.3e}")
        assert wf[b].shape == wav.shape and err <= WAV_TOL_F32, (b, err)


def test_config5_long_form_streaming_full_size(case):
    """BASELINE config 5 at full size: ONE 60 s utterance (1500 decode steps, 24-layer LM, captured decode graph), streamed
    in 25-token hops with the estimator state cache (bf16).
    (a) the concatenated chunks equal one offline DAC decode of the concatenated latents (exact DAC context per hop);
    (b) the cached schedule equals the reference's schedule, which re-solves the whole prefix at every hop
        (cli/model.py:341-352), to bf16 rounding;
    (c) the fp8 MFMA attention variant: same token ids, waveform within a stated SNR of the bf16 stream."""
    from mmx.pipeline import TtsEngine
    N = 1500
    text = torch.randint(0, 151936, (1, 290), generator=torch.Generator().manual_seed(6)).cuda()
    emb = case["emb"].cuda()

    def run(attn, cache):
        eng = TtsEngine(case["llm_sd"], case["flow_sd"], case["dac_sd"], dtype=1, max_batch=1, max_ctx=2048, attn=attn)
        lats = []
        chunks = [c.reshape(-1) for c in eng.tts_stream(text, emb, seed=1, exact_steps=N, latents_out=lats, cache=cache)]
        n_out = int(eng.llm.state[2, 0])
        toks = eng.llm.out_tokens[0, :n_out].tolist()
        wav, lat = torch.cat(chunks), torch.cat(lats, 0)
        # the streaming passes rendered frames [0, nf): their samples (all but the 8 frames held back for the closing cross-fade)
        # must equal one offline decode of those latents wherever the offline decode has the same right context
        ns = wav.shape[0] - chunks[-1].shape[0]
        nf = ns // 480 + 8
        off = eng.dac.decode_time_major(lat[:nf].to(eng.dac.tdt).reshape(1, -1, 80).contiguous(), 1, nf)[0, 0]
        CR, nh = eng.dac.ctx_right, (n_out - 3) // 25
        del eng
        torch.cuda.empty_cache()
        return wav, lat, off, toks, CR, nh, ns

    wav, lat, off, toks, CR, nh, ns = run("bf16", True)
    assert N - 20 <= len(toks) <= N and wav.shape[0] == len(toks) * 960 and lat.shape == (2 * len(toks), 80)
    ncmp = ns - CR * 480
    err = (wav[:ncmp] - off[:ncmp]).abs().max().item()
    print(f"config 5 (60 s, {len(toks)} ids, {nh + 1} chunks): streamed samples vs offline decode of the same latents {err:.3e}")
    assert err < 2e-2 and torch.isfinite(wav).all()
    wav_r, lat_r, _, toks_r, _, _, _ = run("bf16", False)
    dl, dw = (lat - lat_r).abs().max().item(), (wav - wav_r).abs().max().item()
    snr_r = _snr_db(wav_r, wav)
    print(f"config 5: cached state vs prefix recompute: latents {dl:.3e}, waveform {dw:.3e}, SNR {snr_r:.1f} dB")
    # The two schedules run different kernel variants (16-query hop kernels vs whole-prefix tiles), so every bf16 activation
    # is rounded on a different value.  Each schedule is one bf16 evaluation of the same exact path, i.e. within the bf16
    # build's own distance from it - BF16_MIN_SNR_DB + 10 = 35 dB measured against the oracle at config-3 size
    # (test_composed_pipeline_bf16_bound) - and two such evaluations differ by at most the sum of their errors: 35 - 3 = 32 dB.
    # (The fp32 and split builds agree to 2e-5: tests/test_gpu_stream.py.)
    assert toks_r == toks and snr_r >= 32.0, (dw, snr_r)
    wav_8, _, _, toks_8, _, _, _ = run("fp8", True)
    snr = _snr_db(wav, wav_8)
    print(f"config 5: fp8 attention vs bf16 attention: waveform SNR {snr:.1f} dB, max abs diff {(wav - wav_8).abs().max().item():.3e}")
    # e4m3 attention operands (3-bit mantissa, 2^-4 per operand) against bf16's 2^-9 only inside the 56 x 10 attention products
    # of the estimator: stated 45 dB (measured 53.8 dB in round 2)
    assert toks_8 == toks and torch.isfinite(wav_8).all() and snr >= 45.0, snr


def test_graft_entry_smoke():
    """The driver's smoke(): reduced-depth LM / estimator through the whole path vs the CPU oracle (fp32 ids identical,
    waveform <= 1e-3), then the bf16 build."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").smoke()
