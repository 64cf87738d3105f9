"""Block-level drop-in classes (speech/matcha/models/components/{decoder,transformer,flow_matching}.py,
speech/cosyvoice/flow/decoder.py:36-85, Qwen2Encoder.forward_one_step, Qwen2LM.inference_wrapper,
ConditionalCFM.solve_euler / forward_estimator) on the HIP kernels vs the reference's own outputs for every sub-block
(tests/golden/blocks.npz, written by oracle/gen_golden.py from the reference classes)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 7
TOL = {0: 2e-4, 1: 6e-2}          # fp32 build abs; bf16 build abs on outputs of std 0.3 .. 1.1 (bf16 operand rounding)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "blocks.npz"))


@pytest.fixture(scope="module")
def flow_sd(golden_dir):
    from oracle import weights as W
    return W.synth_state_dict(W.load_manifest(os.path.join(golden_dir, "manifest_flow.json")), SEED)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _mod(cls, sd, dt, *a, **k):
    m = cls(*a, **k)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda")
    m.compute_dtype = dt
    for s in m.modules():
        if hasattr(s, "compute_dtype"):
            s.compute_dtype = dt
    return m


def _cmp(y, ref, dt, name):
    err = (y.cpu() - torch.from_numpy(ref)).abs().max().item()
    print(f"{name} dtype {dt}: max abs err {err:.3e} (std {ref.std():.3f})")
    assert y.shape == ref.shape and err < TOL[dt], (name, dt, err)


@pytest.mark.parametrize("dt", [0, 1])
def test_estimator_sub_blocks_vs_reference(gold, flow_sd, dt):
    from cosyvoice.flow.decoder import CausalBlock1D, CausalResnetBlock1D
    from matcha.models.components.decoder import TimestepEmbedding
    from matcha.models.components.transformer import BasicTransformerBlock, FeedForward
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    e = "decoder.estimator."
    te = _mod(TimestepEmbedding, _sub(flow_sd, e + "time_mlp."), dt, 320, 1024)
    _cmp(te(t("temb_in")), gold["temb"], dt, "TimestepEmbedding")
    cb = _mod(CausalBlock1D, _sub(flow_sd, e + "final_block."), dt, 256, 256)
    _cmp(cb(t("cb_x"), t("cb_mask")), gold["cb_out"], dt, "CausalBlock1D")
    rb = _mod(CausalResnetBlock1D, _sub(flow_sd, e + "mid_blocks.0.0."), dt, 256, 256, 1024)
    _cmp(rb(t("cb_x"), t("cb_mask"), t("temb")), gold["rb_out"], dt, "CausalResnetBlock1D(256)")
    rb0 = _mod(CausalResnetBlock1D, _sub(flow_sd, e + "down_blocks.0.0."), dt, 320, 256, 1024)
    _cmp(rb0(t("rb0_x"), t("cb_mask"), t("temb")), gold["rb0_out"], dt, "CausalResnetBlock1D(320)")
    tb = _mod(BasicTransformerBlock, _sub(flow_sd, e + "mid_blocks.0.1.0."), dt, 256, 8, 64, dropout=0.0, activation_fn="gelu")
    _cmp(tb(t("tb_hs"), attention_mask=t("tb_bias_pad")), gold["tb_out_pad"], dt, "BasicTransformerBlock(pad mask)")
    _cmp(tb(t("tb_hs"), attention_mask=t("tb_bias_chunk")), gold["tb_out_chunk"], dt, "BasicTransformerBlock(chunk mask)")
    ff = _mod(FeedForward, _sub(flow_sd, e + "mid_blocks.0.1.0.ff."), dt, 256, activation_fn="gelu")
    _cmp(ff(t("tb_hs")), gold["ff_out"], dt, "FeedForward")


@pytest.mark.parametrize("dt", [0, 1])
def test_matcha_groupnorm_blocks_vs_reference(gold, dt):
    from matcha.models.components.decoder import Block1D, ResnetBlock1D
    t = lambda k: torch.from_numpy(gold[k]).cuda()
    w = lambda p: {k[len(p):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(p)}
    mb = _mod(Block1D, w("mb."), dt, 64, 96, groups=8)
    _cmp(mb(t("m_x"), t("cb_mask")), gold["mb_out"], dt, "Block1D")
    mr = _mod(ResnetBlock1D, w("mr."), dt, 64, 96, 128, groups=8)
    _cmp(mr(t("m_x"), t("cb_mask"), t("m_te")), gold["mr_out"], dt, "ResnetBlock1D")


@pytest.mark.parametrize("dt,tol", [(0, 2e-4), (1, 5e-2)])
def test_forward_one_step_vs_oracle(dt, tol):
    """Qwen2Encoder.forward_one_step (llm.py:359-371): prompt rows, then single rows with the returned cache, equal the
    oracle's backbone (the HF restatement pinned to the reference's golden log-probs in tests/test_oracle_golden.py)."""
    from cosyvoice.llm.llm import Qwen2Encoder
    from mmx import shapes, synth
    from oracle import llm as OL
    sd = synth.synth_state_dict(shapes.llm_manifest(layers=2), 1)
    enc = Qwen2Encoder({"num_hidden_layers": 2})
    enc.load_state_dict({k[len("llm."):]: v for k, v in sd.items() if k.startswith("llm.")}, strict=True)
    enc = enc.to("cuda")
    enc.compute_dtype = dt
    g = torch.Generator().manual_seed(2)
    xs = torch.randn(1, 9, 896, generator=g) * 0.05
    cfg = OL.QwenCfg(layers=2)
    y_ref, cache_ref = OL.qwen2_forward(sd, cfg, xs, None)
    y, cache = enc.forward_one_step(xs.cuda(), torch.tril(torch.ones(1, 9, 9)).bool().cuda())
    assert cache.rows == 9 and (y.cpu() - y_ref).abs().max().item() < tol
    for i in range(3):
        x1 = torch.randn(1, 1, 896, generator=g) * 0.05
        y_ref, cache_ref = OL.qwen2_forward(sd, cfg, x1, cache_ref)
        y, cache = enc.forward_one_step(x1.cuda(), torch.ones(1, 1, 1).bool().cuda(), cache)
        assert (y.cpu() - y_ref).abs().max().item() < tol, i
    assert cache.rows == 12


def test_inference_wrapper_equals_inference():
    from test_dropin_api import build_llm
    from mmx import shapes, synth
    lm = build_llm(2)
    lm.load_state_dict(synth.synth_state_dict(shapes.llm_manifest(layers=2), 0), strict=True)
    lm = lm.to("cuda")
    lm.seed = 4
    text = torch.randint(0, 151936, (1, 7), generator=torch.Generator().manual_seed(1)).cuda()
    z = torch.zeros(1, 0, dtype=torch.int64, device="cuda")
    i32 = lambda n: torch.tensor([n], dtype=torch.int32, device="cuda")
    a = list(lm.inference(text=text, text_len=i32(7), prompt_text=z, prompt_text_len=i32(0), prompt_speech_token=z,
                          prompt_speech_token_len=i32(0), embedding=torch.zeros(1, 192, device="cuda")))
    x = lm.engine(1).build_lm_input(text, z, z).unsqueeze(0)
    b = list(lm.inference_wrapper(x, 25, 14, 140, "u"))
    assert a == b and 14 <= len(a) <= 140


def test_solve_euler_equals_forward(golden_dir, flow_sd):
    """ConditionalCFM.solve_euler + forward_estimator (flow_matching.py:74-131, one estimator call per step through the
    module seam) give the latents of CausalConditionalCFM.forward (the recorded whole-solve graph)."""
    from test_dropin_api import build_flow
    flow = build_flow()
    flow.load_state_dict(flow_sd, strict=True)
    flow = flow.to("cuda").float_parity()
    cfm = flow.decoder
    g = torch.Generator().manual_seed(6)
    T = 44
    mu, cond = torch.randn(1, 80, T, generator=g).cuda(), torch.zeros(1, 80, T).cuda()
    spks, mask = torch.randn(1, 80, generator=g).cuda(), torch.ones(1, 1, T).cuda()
    ref, _ = cfm(mu, mask, 10, spks=spks, cond=cond)
    t_span = 1 - torch.cos(torch.linspace(0, 1, 11) * 0.5 * torch.pi)
    z = cfm.rand_noise[:, :, :T].cuda()
    got = cfm.solve_euler(z, t_span=t_span.cuda(), mu=mu, mask=mask, spks=spks, cond=cond)
    assert (got - ref).abs().max().item() < 2e-4
