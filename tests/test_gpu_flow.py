"""Flow decoder (encoder + estimator + 10-step CFM) on the HIP path vs the golden vectors produced by the
reference's own CausalMaskedDiffWithXvec (tests/golden/flow.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 7
# fp32 build: exact-fp32 MFMA, only summation order differs.  bf16 build: bf16 weights/GEMM inputs through 70
# blocks x 10 Euler steps; the bound is what this test measures and pins, relative to outputs of std ~1.2.
EST_TOL = {0: 2e-4, 1: 6e-2}
# flow.inference = 10 Euler steps of the estimator with classifier-free guidance: x += dt * ((1 + cfg) d_cond - cfg d_uncond),
# sum dt = 1, cfg = 0.7, so an estimator error e per call gives at most (1 + 2 cfg) e = 2.4 e on the latents:
# bf16 2.4 x EST_TOL = 0.144 (measured 1.5e-2), fp32 2.4 x 2e-4 = 4.8e-4
FLOW_TOL = {0: 4.8e-4, 1: 1.44e-1}


@pytest.fixture(scope="module")
def engines(golden_dir):
    from mmx.flow import FlowEngine
    from oracle import weights as W
    sd = W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_flow.json")), SEED)
    return {dt: FlowEngine(sd, dtype=dt) for dt in (0, 1)}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "flow.npz"))


@pytest.mark.parametrize("dt", [0, 1])
def test_estimator_seam(engines, gold, dt):
    eng = engines[dt]
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    T = 64
    ones = torch.ones(2, 1, T).cuda()
    for name, streaming, mk in (("est_full", False, ones), ("est_stream", True, ones), ("est_padmask", False, t("est_mask2"))):
        y = eng.estimator_channels_first(t("est_x"), mk, t("est_mu"), t("est_t"), t("est_spks"), t("est_cond"), streaming)
        ref = t(name)
        err = (y - ref).abs().max().item()
        print(f"estimator seam {name} dtype {dt}: max abs err {err:.3e} (std {ref.std().item():.2f})")
        assert err < EST_TOL[dt], (name, dt, err)


def snr_db(ref, got):
    return float(10 * torch.log10(ref.pow(2).mean() / (ref - got).pow(2).mean().clamp_min(1e-30)))


# encoder output (after after_norm, std ~1): fp32 build abs bound; bf16 build a stated SNR requirement
ENC_TOL_F32, ENC_MIN_SNR_BF16 = 2e-4, 30.0


@pytest.mark.parametrize("dt", [0, 1])
def test_encoder_vs_reference_golden(engines, gold, dt):
    """UpsampleConformerEncoder on the HIP path vs the reference's own outputs (tests/golden/flow.npz: `enc_full` =
    25 embedded tokens, no context, full attention; `enc_ctx_stream` = the same with 3 look-ahead context rows and
    chunk-causal masks), upsample_encoder.py:243-316."""
    eng = engines[dt]
    xs, ctx = torch.from_numpy(gold["enc_xs"]).cuda()[0], torch.from_numpy(gold["enc_ctx"]).cuda()[0]
    cases = (("enc_full", xs, True, False), ("enc_ctx_stream", torch.cat([xs, ctx], 0), False, True))
    for name, rows, finalize, streaming in cases:
        h = eng.encode_embedded(rows.to(eng.tdt).contiguous(), finalize, streaming, hidden=True)
        ref = torch.from_numpy(gold[name]).cuda()[0]
        assert h.shape == ref.shape == (50, 512)
        err, snr = (h - ref).abs().max().item(), snr_db(ref, h)
        print(f"encoder {name} dtype {dt}: max abs err {err:.3e}, SNR {snr:.1f} dB")
        if dt == 0:
            assert err < ENC_TOL_F32, (name, err)
        else:
            assert snr > ENC_MIN_SNR_BF16, (name, snr, err)


def test_dropin_encoder_forward_vs_reference_golden(gold, golden_dir):
    """The drop-in UpsampleConformerEncoder.forward(xs, xs_lens, context, streaming) (embedding-in entry point)."""
    from test_dropin_api import build_flow
    from oracle import weights as W
    flow = build_flow()
    flow.load_state_dict(W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_flow.json")), SEED), strict=True)
    enc = flow.encoder.to("cuda").float_parity()
    xs, ctx = torch.from_numpy(gold["enc_xs"]).cuda(), torch.from_numpy(gold["enc_ctx"]).cuda()
    h, m = enc(xs, torch.tensor([25]).cuda())
    assert m.shape == (1, 1, 50) and (h.cpu() - torch.from_numpy(gold["enc_full"])).abs().max().item() < ENC_TOL_F32
    h, _ = enc(xs, torch.tensor([25]).cuda(), context=ctx, streaming=True)
    assert (h.cpu() - torch.from_numpy(gold["enc_ctx_stream"])).abs().max().item() < ENC_TOL_F32


def test_capture_while_other_thread_decodes(engines):
    """The concurrency rule of mmx/flow.py: a thread records a new Euler-solve graph (thread-local capture) while
    another thread keeps replaying its own recorded decode graph on its own stream; both results stay correct."""
    import threading
    from mmx import shapes, synth
    from mmx.llm import LlmEngine
    sd = synth.synth_state_dict(shapes.llm_manifest(layers=2), 0)
    z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
    text = torch.randint(0, 151936, (1, 9), generator=torch.Generator().manual_seed(1)).cuda()
    lm = LlmEngine(sd, dtype=1, max_batch=1, max_ctx=512)
    x = lm.build_lm_input(text, z, z)
    lm.start([x], [300], [300], seed=1)
    ref = lm.run(300)[0]                                   # eager step + capture happen here, single threaded
    eng = engines[1]
    g = torch.Generator().manual_seed(23)
    mus = {T: torch.randn(T, 80, generator=g).cuda() for T in (36, 44, 52, 60)}
    spk, errs, out = torch.randn(80, generator=g).cuda(), [], {}
    base = {T: eng.cfm(mus[T], spk, torch.zeros(T, 80).cuda()).clone() for T in mus}     # eager pass of every shape
    torch.cuda.synchronize()
    go, stop = threading.Event(), threading.Event()

    def decode_thread():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                lm.start([x], [300], [300], seed=1)
                go.set()
                out["toks"] = lm.run(300)[0]               # graph replays while the main thread captures
        except BaseException as e:
            errs.append(e)
        finally:
            go.set()
            stop.set()

    th = threading.Thread(target=decode_thread)
    th.start()
    go.wait()
    with torch.cuda.stream(torch.cuda.Stream()):
        for T in mus:                                      # second call of each shape = hipGraph capture
            y = eng.cfm(mus[T], spk, torch.zeros(T, 80).cuda())
            torch.cuda.current_stream().synchronize()
            assert (y - base[T]).abs().max().item() < 1e-5, T
    th.join()
    assert not errs, errs
    assert out["toks"] == ref


@pytest.mark.parametrize("dt", [0, 1])
def test_flow_inference_vs_reference_golden(engines, gold, dt):
    eng = engines[dt]
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    tok, ptok, pfeat, emb = t("flow_tok"), t("flow_ptok"), t("flow_pfeat"), t("flow_emb")
    none_tok, none_feat = torch.zeros(1, 0, dtype=torch.long).cuda(), torch.zeros(1, 0, 80).cuda()
    cases = {
        "flow_noprompt": (tok, none_tok, none_feat, False, True),
        "flow_prompt": (tok, ptok, pfeat, False, True),
        "flow_stream_nofinal": (tok, ptok, pfeat, True, False),
        "flow_stream_final": (tok, ptok, pfeat, True, True),
    }
    for rep in range(3):             # eager warm-up, graph capture, graph replay must all agree
        for name, (tk, pt, pf, streaming, finalize) in cases.items():
            y = eng.inference(tk, pt, pf, emb, streaming, finalize)
            ref = t(name)
            assert y.shape == ref.shape, (name, y.shape, ref.shape)
            err = (y - ref).abs().max().item()
            if rep == 0:
                print(f"flow.inference {name} dtype {dt}: max abs err {err:.3e} (std {ref.std().item():.2f})")
            assert err < FLOW_TOL[dt], (name, dt, rep, err)


@pytest.mark.parametrize("dt,tol", [(0, 2e-4), (1, 1.5e-1)])
def test_batched_masked_cfm_equals_single(engines, dt, tol):
    """Three utterances of different length solved in ONE padded + masked batch == three separate solves
    (the padding never leaks into valid frames: causal convs, key masks, row masks)."""
    eng = engines[dt]
    g = torch.Generator().manual_seed(17)
    Ts = [40, 64, 52]
    mus = [torch.randn(t, 80, generator=g).cuda() for t in Ts]
    conds = [torch.zeros(t, 80).cuda() for t in Ts]
    spks = [torch.randn(80, generator=g).cuda() for _ in Ts]
    single = [eng.cfm(m, s, c).clone() for m, s, c in zip(mus, spks, conds)]
    for rep in range(3):
        batch = eng.cfm_batch(mus, spks, conds, pad_to=32)
        for a, b in zip(batch, single):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() < tol, (dt, rep, (a - b).abs().max().item())


@pytest.mark.parametrize("dt,tol", [(0, 2e-5), (1, 6e-2)])
def test_batched_encoder_equals_single(engines, dt, tol):
    """FlowEngine.encode_batch (one zero-padded pass over a flow group) == the per-utterance encoder: the padding never
    reaches a valid row (masked embedding rows for the look-ahead conv, left-looking convs, key lengths in the attention,
    length-independent relative positions).  fp32 build: the same arithmetic per row; bf16 build: the MFMA tiles of a
    row differ between the two launches' shapes only in nothing - the bound is the build's own rounding noise."""
    eng = engines[dt]
    g = torch.Generator().manual_seed(5)
    vocab = eng.emb_table.shape[0]
    ids = [torch.randint(0, vocab, (n,), generator=g).cuda() for n in (37, 64, 9, 50)]
    single = [eng.encode(i, True, False).clone() for i in ids]
    batch = eng.encode_batch(ids)
    for a, b, i in zip(batch, single, ids):
        assert a.shape == b.shape == (2 * i.numel(), 80)
        err = (a - b).abs().max().item()
        assert err < tol, (dt, i.numel(), err)
    cb = eng.conditions_batch([i[3:].reshape(1, -1) for i in ids], [i[:3].reshape(1, -1) for i in ids],
                              [torch.randn(1, 6, 80, generator=g).cuda() for _ in ids], [torch.randn(1, 192, generator=g).cuda() for _ in ids])
    assert [c[0].shape[0] for c in cb] == [2 * i.numel() for i in ids] and all(c[3] == 6 and c[1].shape == (1, 80) for c in cb)


def test_estimator_max_profile_length_bf16_vs_fp32(engines):
    """T = 3000 frames (the reference's TensorRT profile maximum, cli/model.py:96-101), ragged (not a multiple of any
    tile): the bf16 build (MFMA flash attention) agrees with the fp32 build (dense attention) within the bf16 bound."""
    T = 3000
    g = torch.Generator().manual_seed(2)
    x, mu = torch.randn(2, 80, T, generator=g).cuda(), torch.randn(2, 80, T, generator=g).cuda()
    cond, spks = torch.zeros(2, 80, T).cuda(), torch.randn(2, 80, generator=g).cuda()
    t, mask = torch.tensor([0.5, 0.5]).cuda(), torch.ones(2, 1, T).cuda()
    a = engines[0].estimator_channels_first(x, mask, mu, t, spks, cond, True)
    b = engines[1].estimator_channels_first(x, mask, mu, t, spks, cond, True)
    assert a.shape == (2, 80, T) and torch.isfinite(a).all() and torch.isfinite(b).all()
    assert (a - b).abs().max().item() < 0.15 and (a - b).abs().mean().item() < 0.01


def test_capture_survives_unreachable_engine_and_eager_collector():
    """Regression for the process abort of round 2 (a recorded plan is a reference cycle; an engine dropped without
    close() leaves its hipGraphs to Python's cyclic collector, which used to run INSIDE another engine's capture and
    destroy them on the capturing thread).  An unreachable engine is left uncollected, the collector is made as eager as
    it gets, and a new plan is captured: it must complete.  Also: eviction and close() destroy graphs deterministically."""
    import gc
    from mmx import shapes, synth
    from mmx.flow import FlowEngine
    sd = synth.synth_state_dict(shapes.flow_manifest(num_mid_blocks=1), 3)
    mk = lambda: FlowEngine(sd, dtype=1, parts=("estimator",))
    mu, cond, spk = torch.randn(64, 80).cuda(), torch.zeros(64, 80).cuda(), torch.randn(80).cuda()
    gc.collect()
    gc.disable()
    try:
        e = mk()
        e.cfm(mu, spk, cond)
        e.cfm(mu, spk, cond)                                 # second call records the plan's graph
        del e                                                # unreachable, NOT collected: its plan is a cycle
    finally:
        gc.enable()
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        e2 = mk()
        e2.cfm(mu, spk, cond)
        x = e2.cfm(mu, spk, cond).clone()                    # captures with the collector at its most eager
    finally:
        gc.set_threshold(*old)
    assert torch.isfinite(x).all()
    # eviction releases the evicted plan's graph on the spot and the accounting follows
    e2.plan_budget_bytes = 1
    p_old = next(iter(e2._plans.values()))
    run_old = p_old.run
    assert run_old.graph is not None
    e2.cfm(mu[:32], spk, cond[:32])
    assert run_old.graph is None and run_old.fn is None and len(e2._plans) == 1
    e2.close()
    assert e2.plan_bytes == 0 and not e2._plans
