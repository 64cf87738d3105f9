"""AR speech-token LM on the HIP path vs the golden vectors produced by the reference's Qwen2LM modules
(full-size 24-layer backbone, tests/golden/llm.npz) and vs the CPU oracle's free-running decode."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 7


@pytest.fixture(scope="module")
def llm_sd(golden_dir):
    from oracle import weights as W
    return W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_llm.json")), SEED)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "llm.npz"))


@pytest.mark.parametrize("dt,tol", [(0, 2e-3), (1, 0.35)])
def test_teacher_forced_logp_vs_reference_golden(llm_sd, gold, dt, tol):
    """prefill (28 rows) + 16 teacher-forced decode steps: log-probabilities of every step vs the reference.
    fp32 build: 2e-3 abs on log-probs; bf16 build: bf16 weight rounding through 24 layers (bound pinned here)."""
    from mmx.llm import LlmEngine
    eng = LlmEngine(llm_sd, dtype=dt, max_batch=1, max_ctx=256)
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    x = eng.build_lm_input(t("text"), t("ptext"), t("pspeech"))
    assert (x - t("lm_input")[0]).abs().max().item() < 1e-6
    forced = t("forced").reshape(1, -1)
    eng.start([x], [100], [100], seed=5, forced=forced, want_logp=True)
    errs = [(eng.logp[0] - t("logp")[0]).abs().max().item()]
    for i in range(16):
        eng.step()
        errs.append((eng.logp[0] - t("logp")[i + 1]).abs().max().item())
    assert max(errs) < tol, errs
    print(f"dtype {dt}: max |dlogp| over 17 steps = {max(errs):.3e}")
    # teacher forcing: the accepted history is the forced one (17th step reads the zero padding of `forced`)
    assert eng.tokens()[0][:16] == gold["forced"].tolist()


def test_free_running_token_ids_match_oracle_fp32(llm_sd):
    """North-star: FSQ token ids bit-exact vs the CPU path on identical inputs (same Philox noise)."""
    from mmx.llm import LlmEngine
    from oracle import llm as OL
    g = torch.Generator().manual_seed(3)
    text = torch.randint(0, 151936, (1, 10), generator=g)
    ptext = torch.randint(0, 151936, (1, 4), generator=g)
    pspeech = torch.randint(0, 6561, (1, 6), generator=g)
    nsteps = 24
    want = OL.lm_inference(llm_sd, OL.QwenCfg(), text, ptext, pspeech, seed=11, seq=0, max_steps=nsteps)
    eng = LlmEngine(llm_sd, dtype=0, max_batch=1, max_ctx=256)
    x = eng.build_lm_input(text.cuda(), ptext.cuda(), pspeech.cuda())
    eng.start([x], [int(10 * 2)], [int(10 * 20)], seed=11)
    got = eng.run(nsteps)[0]
    assert got == want, (got, want)


def test_batched_decode_matches_single(llm_sd):
    """Batch of 3 sequences with different prompts == three single-sequence runs (bf16, same seeds/seq ids)."""
    from mmx.llm import LlmEngine
    g = torch.Generator().manual_seed(9)
    reqs = []
    for L in (7, 19, 12):
        reqs.append((torch.randint(0, 151936, (1, L), generator=g).cuda(), torch.zeros(1, 0, dtype=torch.long).cuda(),
                     torch.randint(0, 6561, (1, 5), generator=g).cuda()))
    single = []
    e1 = LlmEngine(llm_sd, dtype=1, max_batch=1, max_ctx=256)
    for i, r in enumerate(reqs):
        e1.start([e1.build_lm_input(*r)], [12], [12], seed=2, seq_ids=[i])
        single.append(e1.run(12)[0])
    e3 = LlmEngine(llm_sd, dtype=1, max_batch=3, max_ctx=256)
    e3.start([e3.build_lm_input(*r) for r in reqs], [12] * 3, [12] * 3, seed=2, seq_ids=[0, 1, 2])
    got = e3.run(12)
    assert got == single


def test_max_len_stops_sequences_independently(llm_sd):
    """exact-length decode (min_len == max_len, EOS ignored): every sequence stops at its own max_len
    (llm.py:746 `for i in range(max_len)`), tokens <= steps (ids > EOS are skipped, llm.py:755)."""
    from mmx.llm import LlmEngine, ST_STEP, ST_FIN
    g = torch.Generator().manual_seed(4)
    eng = LlmEngine(llm_sd, dtype=1, max_batch=3, max_ctx=256)
    z = torch.zeros(1, 0, dtype=torch.long).cuda()
    xs = [eng.build_lm_input(torch.randint(0, 151936, (1, 9), generator=g).cuda(), z, z) for _ in range(3)]
    lens = [5, 17, 11]
    eng.start(xs, lens, lens, seed=1)
    toks = eng.run(max(lens))
    assert eng.state[ST_FIN].tolist() == [1, 1, 1]
    assert eng.state[ST_STEP].tolist() == lens
    assert all(len(t) <= n and len(t) >= n - 2 for t, n in zip(toks, lens))


def test_overlapped_batch_pipeline_equals_sequential(golden_dir):
    """tts_batch with the LM decode and the flow/DAC stage overlapped on two streams/threads returns the same
    waveforms as the back-to-back schedule (reduced-depth models, 6 utterances of different length)."""
    from mmx import shapes, synth
    from mmx.pipeline import TtsEngine
    llm_sd = synth.synth_state_dict(shapes.llm_manifest(layers=2, vocab=4096), 0)
    flow_sd = synth.synth_state_dict(shapes.flow_manifest(num_mid_blocks=1), 0)
    dac_sd = synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0)
    eng = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=1, max_batch=6, max_ctx=256)
    g = torch.Generator().manual_seed(0)
    texts = [torch.randint(0, 4096, (1, 8), generator=g).cuda() for _ in range(6)]
    emb = [torch.randn(1, 192, generator=g).cuda() for _ in range(6)]
    lens = [9, 30, 17, 30, 12, 24]
    ref = eng.tts_batch(texts, emb, seed=3, exact_steps=lens, group_size=2, overlap=False)
    ref = [w.clone() for w in ref]
    for rep in range(3):       # eager warm-up, capture, replay
        got = eng.tts_batch(texts, emb, seed=3, exact_steps=lens, group_size=2, overlap=True)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert a.shape == b.shape and (a - b).abs().max().item() < 2e-2, (rep, (a - b).abs().max().item())


def test_compaction_to_smaller_batch_preserves_tokens(llm_sd):
    """A 20-sequence batch that continues in the 16-slot engine once <= 16 sequences are active produces the same
    tokens as the same batch decoded without compaction (same weights, same KV pages, same Philox streams)."""
    from mmx.llm import LlmEngine, ST_FIN, ST_NOUT
    g = torch.Generator().manual_seed(8)
    z = torch.zeros(1, 0, dtype=torch.long).cuda()
    B = 20
    lens = [6 + (i * 3) % 17 for i in range(B)]
    big = LlmEngine(llm_sd, dtype=1, max_batch=B, max_ctx=128)
    small = LlmEngine(None, dtype=1, max_batch=16, max_ctx=128, share_from=big)
    texts = [torch.randint(0, 151936, (1, 5), generator=g).cuda() for _ in range(B)]
    xs = [big.build_lm_input(t, z, z) for t in texts]
    big.start(xs, lens, lens, seed=4)
    ref = big.run(max(lens))
    big.start(xs, lens, lens, seed=4)
    out = [None] * B
    eng, slots, done = big, list(range(B)), 1
    while done < max(lens):
        eng.step()
        done += 1
        fin, n = eng.state[ST_FIN].tolist(), eng.state[ST_NOUT].tolist()
        for s_, b in enumerate(slots):
            if fin[s_] and out[b] is None:
                out[b] = eng.out_tokens[s_, :n[s_]].tolist()
        active = [s_ for s_, b in enumerate(slots) if out[b] is None]
        if eng is big and 0 < len(active) <= 16:
            small.compact_from(big, active)
            eng, slots = small, [slots[s_] for s_ in active]
    n = eng.state[ST_NOUT].tolist()
    for s_, b in enumerate(slots):
        if out[b] is None:
            out[b] = eng.out_tokens[s_, :n[s_]].tolist()
    assert out == ref


def test_inference_spk_lm_input_matches_oracle(llm_sd):
    """llm.py:634-665: [sos | spk_embed_affine(normalize(e)) | text | task_id | prompt speech] (fp32 build)."""
    from mmx import ops
    from mmx.llm import LlmEngine
    from oracle import spk as OSPK
    import torch.nn.functional as F
    eng = LlmEngine(llm_sd, dtype=0, max_batch=1, max_ctx=128)
    g = torch.Generator().manual_seed(6)
    text, ptext = torch.randint(0, 151936, (1, 7), generator=g), torch.randint(0, 151936, (1, 3), generator=g)
    pspeech = torch.randint(0, 6561, (1, 4), generator=g)
    e = torch.randn(1, 192, generator=g)
    w, b = llm_sd["spk_embed_affine_layer.weight"], llm_sd["spk_embed_affine_layer.bias"]
    spk = eng.speaker_conditioning(ops.pack_linear(w.cuda(), 0), b.cuda().contiguous(), e.cuda())
    x = eng.build_lm_input(text.cuda(), ptext.cuda(), pspeech.cuda(), speaker_embed=spk)
    ref = OSPK.build_lm_input_spk(llm_sd, text, ptext, pspeech, F.linear(F.normalize(e, dim=1), w, b).unsqueeze(1))
    assert x.shape == ref.shape[1:] and (x.cpu() - ref[0]).abs().max().item() < 1e-5


@pytest.mark.parametrize("dt,tol", [(0, 2e-3), (1, 0.25)])
def test_batched_prefill_matches_per_sequence_prefill(dt, tol):
    """LlmEngine._prefill_batch (all prompts in one tall-GEMM pass, ragged lengths zero padded) against the
    per-sequence prefill through the decode kernels: same first-step logits, and in the fp32 build the same tokens."""
    from mmx import shapes, synth
    from mmx.llm import LlmEngine
    sd = synth.synth_state_dict(shapes.llm_manifest(layers=2), 0)
    g = torch.Generator().manual_seed(11)
    B = 5
    texts = [torch.randint(0, 151936, (1, n), generator=g).cuda() for n in (7, 30, 12, 70, 3)]
    z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
    outs = []
    for batched in (True, False):
        eng = LlmEngine(sd, dtype=dt, max_batch=B, max_ctx=256)
        if not batched:
            eng.pf_layers = []
        xs = [eng.build_lm_input(t, z, z) for t in texts]
        eng.start(xs, [20] * B, [20] * B, seed=5)
        logits = eng.logits.clone()
        toks = eng.run(20)
        outs.append((logits, toks))
    err = (outs[0][0] - outs[1][0]).abs().max().item()
    assert err < tol, err
    if dt == 0:
        assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize("dt", [0, 1])
def test_continuous_batching_equals_per_request_decode(llm_sd, dt):
    """§8e "continuous batching": 9 requests through a 4-slot engine whose KV pool holds fewer pages than 9 sequences need
    at once — finished sequences return their pages to the allocator and queued requests are admitted into the freed
    slots between decode steps.  fp32: every request's tokens equal its stand-alone single-sequence decode under the
    same seed and sequence id.  bf16: see below."""
    from mmx.llm import LlmEngine
    g = torch.Generator().manual_seed(21)
    z = torch.zeros(1, 0, dtype=torch.long).cuda()
    lens = [9, 31, 14, 40, 7, 22, 35, 12, 18]
    texts = [torch.randint(0, 151936, (1, 4 + i % 5), generator=g).cuda() for i in range(len(lens))]
    want = []
    if dt == 0:
        e1 = LlmEngine(llm_sd, dtype=0, max_batch=1, max_ctx=128)
        for i, (t, n) in enumerate(zip(texts, lens)):
            e1.start([e1.build_lm_input(t, z, z)], [n], [n], seed=6, seq_ids=[i])
            want.append(e1.run(n)[0])
    else:
        # bf16: a request admitted into a running batch is prefilled by the chunked decode kernels and draws its first
        # token inside the batch's decode step; a fixed-batch start() uses the tall-GEMM prompt pass.  Same math, other
        # rounding - so the bf16 statement is scheduling invariance: the same queue on an engine with ample pages, polled
        # every step (other admission times, other neighbours in the batch) gives the same ids per request.
        ea = LlmEngine(llm_sd, dtype=1, max_batch=4, max_ctx=128)
        want = ea.run_queue([(ea.build_lm_input(t, z, z), n, n) for t, n in zip(texts, lens)], seed=6, poll_every=1, ahead=40)
        assert all(len(w) <= n and len(w) >= n - 2 for w, n in zip(want, lens))
    eng = LlmEngine(llm_sd, dtype=dt, max_batch=4, max_ctx=128, kv_pages=14)     # 14 pages of 16 rows: < 9 x 4 pages
    reqs = [(eng.build_lm_input(t, z, z), n, n) for t, n in zip(texts, lens)]
    for rep in range(2):                                                         # second pass: recorded graphs, reused pages
        got = eng.run_queue(reqs, seed=6, poll_every=4, ahead=8)
        assert got == want, rep
        assert eng.pages.n_free == 14 and all(not p for p in eng.slot_pages)


def test_tts_batch_with_more_utterances_than_slots():
    """tts_batch on a 3-slot engine with 7 utterances (the rest queue and are admitted as slots free) returns the
    waveforms of the 7-slot run (fp32 build, reduced-depth models): same ids, same audio."""
    from mmx import shapes, synth
    from mmx.pipeline import TtsEngine
    llm_sd = synth.synth_state_dict(shapes.llm_manifest(layers=2, vocab=4096), 0)
    flow_sd = synth.synth_state_dict(shapes.flow_manifest(num_mid_blocks=1), 0)
    dac_sd = synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0)
    g = torch.Generator().manual_seed(2)
    texts = [torch.randint(0, 4096, (1, 8), generator=g).cuda() for _ in range(7)]
    emb = [torch.randn(1, 192, generator=g).cuda() for _ in range(7)]
    # the ADMITTED utterances (index >= 3) run for 100+ steps: far past the 32 rows an admission used to reserve, so a
    # sequence that decodes into the shared scratch page shows up as different ids
    lens = [12, 30, 9, 121, 16, 140, 104]
    e7 = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=0, max_batch=7, max_ctx=256)
    ref = [w.clone() for w in e7.tts_batch(texts, emb, seed=3, exact_steps=lens, group_size=2)]
    want = [t.tolist() for t in e7.last_tokens]
    e3 = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=0, max_batch=3, max_ctx=256)
    for rep in range(2):
        got = e3.tts_batch(texts, emb, seed=3, exact_steps=lens, group_size=2, poll_every=4)
        torch.cuda.synchronize()
        assert [t.tolist() for t in e3.last_tokens] == want, rep
        for a, b in zip(got, ref):
            assert a.shape == b.shape and (a - b).abs().max().item() < 1e-4, (rep, a.shape, b.shape)
    # a single decode slot: admissions happen only at polls, the loop must still see every utterance
    e1 = TtsEngine(llm_sd, flow_sd, dac_sd, dtype=0, max_batch=1, max_ctx=256)
    got = e1.tts_batch(texts[:3], emb[:3], seed=3, exact_steps=lens[:3], group_size=2, poll_every=4)
    torch.cuda.synchronize()
    assert [t.tolist() for t in e1.last_tokens] == want[:3]
