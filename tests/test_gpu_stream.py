"""Streaming synthesis (BASELINE config 5 at test size; speech/cosyvoice/cli/model.py:336-386): the hop schedule, the
exact DAC context handling (the chunks equal the offline decode of the same latents), the look-ahead logic of the
decode loop, and the reference-shaped CosyVoice2Model.tts(stream=True)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
HOP = 480


@pytest.fixture(scope="module")
def weights():
    from mmx import shapes, synth
    return (synth.synth_state_dict(shapes.llm_manifest(layers=2), 0), synth.synth_state_dict(shapes.flow_manifest(), 0),
            synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0))


def _inputs():
    text = torch.randint(0, 151936, (1, 20), generator=torch.Generator().manual_seed(4)).cuda()
    emb = torch.randn(1, 192, generator=torch.Generator().manual_seed(5)).cuda()
    return text, emb


# the window of a hop has the full receptive field of the DAC decoder on both sides, so the arithmetic per sample is
# the same as in the offline decode: fp32 build to rounding, bf16 build to the rounding of the bf16 activations
STREAM_TOL = {0: 2e-5, 1: 2e-2, 2: 2e-5}
MC = 8                                                     # frames a streaming pass holds back (cli/model.py:258)


@pytest.mark.parametrize("dt", [0, 1, 2])
def test_stream_chunks_equal_offline_decode_of_the_same_latents(weights, dt):
    """TtsEngine.tts_stream: chunk sizes follow the hop schedule (right context + the MEL_CACHE frames of the possible
    closing cross-fade held back); up to the closing seam the concatenated chunks equal ONE offline DAC decode of the
    latents they were rendered from, sample for sample.  The closing chunk (different latents for every frame: the closing
    flow pass runs without chunk masks, like the reference's) is checked against the oracle's schedule in
    test_stream_vs_oracle_schedule_with_prompts."""
    from mmx.pipeline import TtsEngine
    eng = TtsEngine(*weights, dtype=dt, max_batch=1, max_ctx=512)
    text, emb = _inputs()
    CR = eng.dac.ctx_right
    assert (eng.dac.ctx_left, CR) == (16, 15)
    for rep in range(2):                                   # eager, then recorded graphs
        lats = []
        chunks = list(eng.tts_stream(text, emb, seed=3, exact_steps=118, latents_out=lats))
        n_out = int(eng.llm.state[2, 0])
        assert 108 <= n_out <= 118                         # ids above the EOS id are skipped (llm.py:755)
        nh = (n_out - 3) // 25                             # full hops (25 tokens + 3 look-ahead available)
        sizes = [c.shape[-1] for c in chunks]
        assert sizes[:nh] == [(50 - CR - MC) * HOP] + [50 * HOP] * (nh - 1), sizes
        assert sum(sizes) == n_out * 2 * HOP and len(chunks) == nh + 1
        lat = torch.cat(lats, 0)
        assert lat.shape == (2 * n_out, 80)
        ns = sum(sizes[:nh])                               # samples emitted by the streaming passes
        nf = ns // HOP + MC                                # frames the streaming passes rendered
        zt = lat[:nf + CR].to(eng.dac.tdt).reshape(1, -1, 80).contiguous()   # + their right context, streaming values
        got = torch.cat([c.reshape(-1) for c in chunks])
        assert torch.isfinite(got).all()
        # the latents behind the streamed samples are the streaming passes' values; their right context too (the closing
        # pass's re-solve of those frames starts MC frames later): decode a prefix of the streamed latents offline
        off = eng.dac.decode_time_major(zt[:, :nf], 1, nf)[0, 0]
        d = (got[:ns - CR * HOP] - off[:ns - CR * HOP]).abs().max().item()
        print(f"stream vs offline dtype {dt} rep {rep}: max abs err {d:.3e} over {ns - CR * HOP} samples")
        assert d < STREAM_TOL[dt], d


def _prompted_inputs():
    g = torch.Generator().manual_seed(12)
    return dict(text=torch.randint(0, 151936, (1, 20), generator=g), prompt_text=torch.randint(0, 151936, (1, 5), generator=g),
                llm_prompt=torch.randint(0, 6561, (1, 30), generator=g), flow_prompt=torch.randint(0, 6561, (1, 30), generator=g),
                feat=torch.randn(1, 60, 80, generator=g) * 0.5, emb=torch.randn(1, 192, generator=g))


@pytest.mark.parametrize("dt", [0, 2])
def test_stream_vs_oracle_schedule_with_prompts(weights, dt):
    """SURVEY 8f-1 against the ORACLE (oracle/stream.py restates cli/model.py:336-378 over oracle.flow / oracle.dac): a
    zero-shot utterance (LM prompt text + prompt speech tokens, flow prompt tokens + prompt latents: 30 prompt tokens ->
    prompt_token_pad 20, first hop 45 tokens) of 130 tokens = 4 streaming hops + the closing pass.  TtsEngine.tts_stream with
    the estimator state cache and with per-hop recompute: the same chunk sizes as the oracle's schedule and every sample
    within 1e-3 (north-star bound), the closing seam included."""
    from mmx.pipeline import TtsEngine
    from oracle import stream as OS
    llm_sd, flow_sd, dac_sd = weights
    I = _prompted_inputs()
    c = lambda t: t.cuda()
    eng = TtsEngine(*weights, dtype=dt, max_batch=1, max_ctx=512)
    outs = {}
    for cache in (True, False):
        chunks = [w.reshape(-1).cpu() for w in eng.tts_stream(c(I["text"]), c(I["emb"]), seed=5, exact_steps=134, cache=cache,
                                                              prompt_text=c(I["prompt_text"]), llm_prompt_speech_token=c(I["llm_prompt"]),
                                                              flow_prompt_speech_token=c(I["flow_prompt"]), prompt_speech_feat=c(I["feat"]))]
        n_out = int(eng.llm.state[2, 0])
        outs[cache] = (chunks, eng.llm.out_tokens[0, :n_out].cpu().to(torch.int64).reshape(1, -1))
    toks = outs[True][1]
    assert torch.equal(toks, outs[False][1]) and 123 <= toks.shape[1] <= 134
    with torch.no_grad():
        want = OS.tts_stream(flow_sd, dac_sd, [5, 4, 4, 3, 2], toks, I["flow_prompt"], I["feat"], I["emb"], eng.dac.ctx_left, eng.dac.ctx_right)
    assert len(OS.hop_schedule(toks.shape[1], 30)) == 5
    for cache in (True, False):
        got = outs[cache][0]
        assert [g.shape[0] for g in got] == [w.shape[0] for w in want], ([g.shape[0] for g in got], [w.shape[0] for w in want])
        errs = [(g - w).abs().max().item() for g, w in zip(got, want)]
        print(f"tts_stream (cache={cache}) dtype {dt} vs oracle schedule: per-chunk max abs err {[f'{e:.2e}' for e in errs]}")
        assert max(errs) <= 1e-3, errs


@pytest.mark.parametrize("dt,tol", [(0, 2e-5), (1, 2e-2), (2, 1e-4)])
def test_stream_cached_state_equals_recompute(weights, dt, tol):
    """Config 5: hops that solve only their new frames from the cached K / V rows and conv inputs of earlier hops
    (FlowEngine.StreamState) give the waveform of the reference's schedule, which re-solves every frame at every hop
    (cli/model.py:341-352).  fp32: equal to rounding; bf16: to the rounding of bf16 activations; split build: to the rounding of
    its 16-bit products - the two schedules launch different forms of the attention kernel (16 or 32 queries per wave, chosen
    from the grid size), whose lazy softmax rescale is decided per wave: measured 2.6e-5, stated 1e-4 (10 x below the bound)."""
    from mmx.pipeline import TtsEngine
    eng = TtsEngine(*weights, dtype=dt, max_batch=1, max_ctx=512)
    text, emb = _inputs()
    la, lb = [], []
    a = torch.cat([c.reshape(-1) for c in eng.tts_stream(text, emb, seed=3, exact_steps=118, latents_out=la, cache=True)])
    b = torch.cat([c.reshape(-1) for c in eng.tts_stream(text, emb, seed=3, exact_steps=118, latents_out=lb, cache=False)])
    dl = (torch.cat(la) - torch.cat(lb)).abs().max().item()
    dw = (a - b).abs().max().item()
    print(f"cached vs recomputed streaming dtype {dt}: latents max abs diff {dl:.3e}, waveform {dw:.3e}")
    assert a.shape == b.shape and dw < tol and dl < 50 * tol, (dl, dw)


def test_stream_tokens_equal_offline_tokens(weights):
    from mmx.pipeline import TtsEngine
    eng = TtsEngine(*weights, dtype=1, max_batch=1, max_ctx=512)
    text, emb = _inputs()
    list(eng.tts_stream(text, emb, seed=3, exact_steps=90))
    n_out = int(eng.llm.state[2, 0])
    toks_stream = eng.llm.out_tokens[0, :n_out].tolist()
    assert toks_stream == eng.generate_tokens([text], seed=3, exact_steps=90)[0].tolist()


def test_stream_makes_progress_when_many_ids_are_skipped(weights):
    """ids above the EOS id advance the step counter but produce no token (llm.py:755-756).  With more than 8
    of them the old step-counted look-ahead stopped issuing decode steps and the generator never returned."""
    import threading
    from mmx.pipeline import TtsEngine
    eng = TtsEngine(*weights, dtype=1, max_batch=1, max_ctx=512)
    text, emb = _inputs()
    steps = 100
    forced = torch.randint(0, 6561, (1, steps), generator=torch.Generator().manual_seed(9))
    forced[0, 5:19] = 6562                                 # 14 skipped ids ahead of the first hop
    forced[0, 40:52] = 6563
    res = {}

    def run():
        with torch.cuda.stream(torch.cuda.Stream()):
            res["chunks"] = [c.clone() for c in eng.tts_stream(text, emb, seed=1, exact_steps=steps, forced=forced.cuda())]

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(120)
    assert not th.is_alive(), "tts_stream did not terminate"
    n_out = int(eng.llm.state[2, 0])
    assert n_out == steps - 26
    assert sum(c.shape[-1] for c in res["chunks"]) == n_out * 2 * HOP


def test_dropin_tts_stream_vs_oracle_and_engine(weights):
    """CosyVoice2Model.tts(stream=True) (drop-in), zero-shot prompts included: the same chunks as the engine-level stream
    (same tokens under the same seed) and, through it, within 1e-3 of the oracle's schedule (oracle/stream.py)."""
    from functools import partial
    from cosyvoice.cli.model import CosyVoice2Model
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM
    from cosyvoice.utils.common import ras_sampling
    from test_dropin_api import build_dac, build_flow
    from mmx.pipeline import TtsEngine
    from oracle import stream as OS
    llm_sd, flow_sd, dac_sd = weights
    lm = Qwen2LM(896, 896, 6561, Qwen2Encoder({"num_hidden_layers": 2}), partial(ras_sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1))
    lm.load_state_dict(llm_sd, strict=True)
    flow, dac = build_flow(), build_dac(80)
    flow.load_state_dict(flow_sd, strict=True)
    dac.load_state_dict(dac_sd, strict=False)
    for m in (lm, flow, dac):
        m.to("cuda").float_parity()
    lm.seed = 5
    model = CosyVoice2Model(lm, flow, dac)
    I = _prompted_inputs()
    N = 134
    ratio = (N + 0.5) / I["text"].shape[1]                  # exactly N decode steps: min == max token / text ratio
    orig = lm.inference
    lm.inference = lambda **kw: orig(**{**kw, "min_token_text_ratio": ratio, "max_token_text_ratio": ratio})
    chunks = [c["tts_speech"].reshape(-1) for c in model.tts(text=I["text"], flow_embedding=I["emb"], llm_embedding=I["emb"],
                                                             prompt_text=I["prompt_text"], llm_prompt_speech_token=I["llm_prompt"],
                                                             flow_prompt_speech_token=I["flow_prompt"], prompt_speech_feat=I["feat"],
                                                             stream=True)]
    c = lambda t: t.cuda()
    eng = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=0, max_batch=1, max_ctx=2048)
    ref = [w.reshape(-1).cpu() for w in eng.tts_stream(c(I["text"]), c(I["emb"]), seed=5, exact_steps=N, prompt_text=c(I["prompt_text"]),
                                                       llm_prompt_speech_token=c(I["llm_prompt"]), flow_prompt_speech_token=c(I["flow_prompt"]),
                                                       prompt_speech_feat=c(I["feat"]), cache=False)]
    assert [x.shape[0] for x in chunks] == [x.shape[0] for x in ref]
    a, b = torch.cat(chunks), torch.cat(ref)
    assert (a - b).abs().max().item() < 2e-5
    n_out = int(eng.llm.state[2, 0])
    toks = eng.llm.out_tokens[0, :n_out].cpu().to(torch.int64).reshape(1, -1)
    with torch.no_grad():
        want = torch.cat(OS.tts_stream(flow_sd, dac_sd, [5, 4, 4, 3, 2], toks, I["flow_prompt"], I["feat"], I["emb"], eng.dac.ctx_left,
                                       eng.dac.ctx_right))
    err = (a - want).abs().max().item()
    print(f"drop-in tts(stream=True) with prompts vs oracle schedule: max abs err {err:.3e}")
    assert a.shape == want.shape and err <= 1e-3, err
