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
STREAM_TOL = {0: 2e-5, 1: 2e-2}


@pytest.mark.parametrize("dt", [0, 1])
def test_stream_chunks_equal_offline_decode_of_the_same_latents(weights, dt):
    """TtsEngine.tts_stream: chunk sizes follow the hop schedule with the right context held back; the concatenated
    chunks equal ONE offline DAC decode of the concatenated latents.  Only the last `ctx_right` frames ahead of the
    closing chunk may differ: like the reference, the closing flow pass runs without chunk masks, so the right context
    those frames were rendered with is not what the closing pass then emits (bounded, checked separately)."""
    from mmx import ops
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
        assert sizes[:nh] == [(50 - CR) * HOP] + [50 * HOP] * (nh - 1), sizes
        assert sum(sizes) == n_out * 2 * HOP and len(chunks) == nh + 1
        lat = torch.cat(lats, 0)
        assert lat.shape == (2 * n_out, 80)
        zt = lat.to(eng.dac.tdt).reshape(1, -1, 80).contiguous()
        off = eng.dac.decode_time_major(zt, 1, lat.shape[0])[0, 0]
        got = torch.cat([c.reshape(-1) for c in chunks])
        edge = (nh * 50 - CR) * HOP                        # first sample of the closing chunk
        d = (got - off).abs()
        lo = edge - CR * HOP
        err_in = max(d[:lo].max().item(), d[edge:].max().item())
        print(f"stream vs offline dtype {dt} rep {rep}: max abs err {err_in:.3e} (closing boundary: {d[lo:edge].max().item():.3e})")
        assert err_in < STREAM_TOL[dt], err_in
        assert d[lo:edge].max().item() < 0.5 and torch.isfinite(got).all()


@pytest.mark.parametrize("dt,tol", [(0, 2e-5), (1, 2e-2)])
def test_stream_cached_state_equals_recompute(weights, dt, tol):
    """Config 5: hops that solve only their new frames from the cached K / V rows and conv inputs of earlier hops
    (FlowEngine.StreamState) give the waveform of the reference's schedule, which re-solves every frame at every hop
    (cli/model.py:341-352).  fp32: equal to rounding; bf16: to the rounding of bf16 activations."""
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


def test_dropin_tts_stream_equals_offline_decode(weights):
    """CosyVoice2Model.tts(stream=True) (drop-in): same tokens as stream=False under the same seed, chunks follow the
    hop schedule, and every chunk except the samples next to the closing boundary equals the engine-level stream."""
    from functools import partial
    from cosyvoice.cli.model import CosyVoice2Model
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM
    from cosyvoice.utils.common import ras_sampling
    from test_dropin_api import build_dac, build_flow
    from mmx.pipeline import TtsEngine
    llm_sd, flow_sd, dac_sd = weights
    lm = Qwen2LM(896, 896, 6561, Qwen2Encoder({"num_hidden_layers": 2}), partial(ras_sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1))
    lm.load_state_dict(llm_sd, strict=True)
    flow, dac = build_flow(), build_dac(80)
    flow.load_state_dict(flow_sd, strict=True)
    dac.load_state_dict(dac_sd, strict=False)
    for m in (lm, flow, dac):
        m.to("cuda").float_parity()
    lm.seed = 3
    model = CosyVoice2Model(lm, flow, dac)
    _, emb = _inputs()
    text = torch.randint(0, 151936, (1, 6), generator=torch.Generator().manual_seed(8)).cuda()   # <= 120 tokens: 4 hops
    chunks = [c["tts_speech"] for c in model.tts(text=text.cpu(), flow_embedding=emb.cpu(), llm_embedding=emb.cpu(), stream=True)]
    eng = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=0, max_batch=1, max_ctx=2048)
    ref = [c.cpu() for c in eng.tts_stream(text, emb, seed=3)]
    assert [c.shape[-1] for c in chunks] == [c.shape[-1] for c in ref]
    a, b = torch.cat([c.reshape(-1) for c in chunks]), torch.cat([c.reshape(-1) for c in ref])
    assert (a - b).abs().max().item() < 2e-5
