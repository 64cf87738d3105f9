"""oracle/gen_golden.py — TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER only.

Generates the golden vectors under tests/golden/ by importing the reference's own Python classes from
/root/reference (through oracle/ref_shims.py), loading the deterministic synthetic weights of
oracle/weights.py into them, and recording inputs + outputs.  Only data is written: no reference source
or bytecode enters the repo (sys.dont_write_bytecode is set by ref_shims).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [dac] [dacfp32] [dacenc] [flow] [blocks] [llm] [bistream] [sampler] [spk] [stream]
"""
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shims as R  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 7
torch.set_grad_enabled(False)


def np_(t):
    return t.detach().cpu().numpy()


def gen_dac(kind="bf16"):
    """kind "fp32": the checkpoint kind of mmx/synth.py with trained-like weight norms (weight_g = s * ||weight_v||, per-channel
    s that is no power of two; folded weights are general fp32 values) -> dac80_fp32.npz.  SURVEY 8c(i): "default init and a
    scaled variant"."""
    m = R.import_dac()
    for lat in ((80, 128) if kind == "bf16" else (80,)):
        torch.manual_seed(0)
        d = m.DACVAE(encoder_dim=64, encoder_rates=[2, 3, 4, 4, 5], latent_dim=lat, decoder_dim=1536,
                     decoder_rates=[5, 4, 4, 3, 2], sample_rate=24000, d_in=1, d_out=1, weight_init="xavier",
                     activation="snake", gain=1.0).eval()
        sd = {k: v for k, v in d.state_dict().items() if k.startswith("decoder.") or k.startswith("de_conv_pre.")}
        if kind == "bf16":
            W.save_manifest(sd, os.path.join(GOLD, f"manifest_dac{lat}.json"))
        syn = W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED, kind=kind)
        if kind == "fp32":                                  # the fixture is about g != ||v||: make sure it is so
            k0 = "decoder.model.1.block.1."                 # the first ConvTranspose1d: weight_g is [Cin, 1, 1]
            r = syn[k0 + "weight_g"].flatten() / torch.norm_except_dim(syn[k0 + "weight_v"], 2, 0).flatten()
            assert syn[k0 + "weight_g"].shape[0] == syn[k0 + "weight_v"].shape[0] and 0.59 < r.min() < 0.7 and 1.4 < r.max() < 1.51
        missing, unexpected = d.load_state_dict(syn, strict=False)
        assert not unexpected
        out = {}
        for T in (8, 50):
            z = torch.randn(1, lat, T, generator=torch.Generator().manual_seed(100 + T))
            y = d.decode(z)
            out[f"z_T{T}"] = np_(z)
            out[f"wav_T{T}"] = np_(y)
            print(f"dac{lat} T={T}: wav {tuple(y.shape)} absmax {y.abs().max():.4f} std {y.std():.4f}")
        # per-stage activations of the T=8 case (shape + a strided sample, for debugging kernels)
        z = torch.from_numpy(out["z_T8"])
        h = d.de_conv_pre(z)
        out["pre_T8"] = np_(h)
        for i, layer in enumerate(d.decoder.model):
            h = layer(h)
            if i < 6:
                print(f"   stage{i}: {tuple(h.shape)} std {h.std():.3f}")
            if i < 3:
                out[f"stage{i}_T8"] = np_(h)
        np.savez_compressed(os.path.join(GOLD, f"dac{lat}.npz" if kind == "bf16" else f"dac{lat}_fp32.npz"), **out)


def gen_dac_enc():
    """DACVAE.encode / forward goldens."""
    import contextlib
    import io
    m = R.import_dac()
    lat = 80
    torch.manual_seed(0)
    d = m.DACVAE(encoder_dim=64, encoder_rates=[2, 3, 4, 4, 5], latent_dim=lat, decoder_dim=1536,
                 decoder_rates=[5, 4, 4, 3, 2], sample_rate=24000, d_in=1, d_out=1, weight_init="xavier",
                 activation="snake", gain=1.0).eval()
    full = d.state_dict()
    sd = {k: v for k, v in full.items() if k.startswith("encoder.") or k.startswith("en_conv_post.")}
    W.save_manifest(sd, os.path.join(GOLD, "manifest_dacenc.json"))
    syn = W.synth_state_dict({k: v.shape for k, v in full.items()}, SEED)
    d.load_state_dict(syn, strict=True)
    out = {}
    for n in (4800, 11000):                                # 10 and ~23 latent frames; 11000 is not a hop multiple
        g = torch.Generator().manual_seed(200 + n)
        t = torch.arange(n) / 24000.0
        wav = (0.3 * torch.sin(2 * math.pi * 220 * t) + 0.1 * torch.randn(n, generator=g)).reshape(1, 1, n)
        x = d.preprocess(wav, 24000)
        with contextlib.redirect_stdout(io.StringIO()):    # encode() prints shapes
            torch.manual_seed(300 + n)
            z, mu, logs = d.encode(x)
            torch.manual_seed(300 + n)
            noise = torch.randn_like(mu)                   # the draw encode() made
            torch.manual_seed(300 + n)
            fw = d(wav, 24000)
        out[f"wav_{n}"] = np_(wav)
        out[f"noise_{n}"] = np_(noise)
        out[f"z_{n}"], out[f"mu_{n}"], out[f"logs_{n}"] = np_(z), np_(mu), np_(logs)
        out[f"recon_{n}"] = np_(fw["audio"])
        if n % 480:                                        # extract_dac_latents.py:20-36 encodes WITHOUT padding
            with contextlib.redirect_stdout(io.StringIO()):
                torch.manual_seed(400 + n)
                zr, mur, logsr = d.encode(torch.clamp(wav, -1.0, 1.0), 24000)
                torch.manual_seed(400 + n)
                out[f"noise_raw_{n}"] = np_(torch.randn_like(mur))
            out[f"z_raw_{n}"], out[f"mu_raw_{n}"], out[f"logs_raw_{n}"] = np_(zr), np_(mur), np_(logsr)
            print(f"   unpadded: z {tuple(zr.shape)}")
        assert torch.equal(fw["z"], z)
        print(f"dacenc n={n}: z {tuple(z.shape)} mu std {mu.std():.3f} logs [{logs.min():.2f},{logs.max():.2f}] "
              f"recon absmax {fw['audio'].abs().max():.3f}")
    h = d.preprocess(torch.from_numpy(out["wav_4800"]), 24000)
    for i, layer in enumerate(d.encoder.block):
        h = layer(h)
        if i < 7:
            print(f"   stage{i}: {tuple(h.shape)} std {h.std():.3f}")
        if i in (0, 1, 5):
            out[f"stage{i}_4800"] = np_(h[..., :160])      # leading frames only: keeps the fixture small
    np.savez_compressed(os.path.join(GOLD, "dacenc.npz"), **out)


def build_flow():
    R.import_cosyvoice()
    from cosyvoice.flow.flow import CausalMaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.decoder import CausalConditionalDecoder
    from cosyvoice.transformer.upsample_encoder import UpsampleConformerEncoder
    from omegaconf import DictConfig
    torch.manual_seed(0)
    # speech/config.yaml:60-116
    enc = UpsampleConformerEncoder(output_size=512, attention_heads=8, linear_units=2048, num_blocks=6,
                                   dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                                   normalize_before=True, input_layer="linear", pos_enc_layer_type="rel_pos_espnet",
                                   selfattention_layer_type="rel_selfattn", input_size=512, use_cnn_module=False,
                                   macaron_style=False, static_chunk_size=25)
    est = CausalConditionalDecoder(in_channels=320, out_channels=80, channels=[256], dropout=0.0,
                                   attention_head_dim=64, n_blocks=4, num_mid_blocks=12, num_heads=8, act_fn="gelu",
                                   static_chunk_size=50, num_decoding_left_chunks=-1)
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80, cfm_params=DictConfig(dict(
        sigma_min=1e-6, solver="euler", t_scheduler="cosine", training_cfg_rate=0.2, inference_cfg_rate=0.7,
        reg_loss_type="l1", use_immiscible=True, immiscible_k=8, use_contrastive_fm=True,
        contrastive_lambda=0.05)), estimator=est)
    flow = CausalMaskedDiffWithXvec(input_size=512, output_size=80, spk_embed_dim=192, output_type="mel",
                                    vocab_size=6561, input_frame_rate=25, only_mask_loss=True, token_latent_ratio=2,
                                    pre_lookahead_len=3, use_speaker_encoder=False, encoder=enc, decoder=cfm).eval()
    return flow


def gen_flow():
    flow = build_flow()
    sd = flow.state_dict()
    W.save_manifest(sd, os.path.join(GOLD, "manifest_flow.json"))
    syn = W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED)
    flow.load_state_dict(syn, strict=True)
    out = {}
    g = torch.Generator().manual_seed(11)
    # (ii) estimator: one call, T=64, CFG pair, both streaming flags, padded second row
    T = 64
    x = torch.randn(2, 80, T, generator=g)
    mu = torch.randn(2, 80, T, generator=g)
    cond = torch.randn(2, 80, T, generator=g) * 0.5
    spks = torch.randn(2, 80, generator=g)
    t = torch.tensor([0.3, 0.3])
    mask = torch.ones(2, 1, T)
    est = flow.decoder.estimator
    for name, streaming, mk in (("est_full", False, mask), ("est_stream", True, mask)):
        y = est(x, mk, mu, t, spks, cond, streaming=streaming)
        out[name] = np_(y)
        print(name, tuple(y.shape), f"std {y.std():.3f} absmax {y.abs().max():.3f}")
    mask2 = mask.clone()
    mask2[1, :, 40:] = 0
    y = est(x, mask2, mu, t, spks, cond, streaming=False)
    out["est_padmask"] = np_(y)
    out.update(est_x=np_(x), est_mu=np_(mu), est_cond=np_(cond), est_spks=np_(spks), est_t=np_(t), est_mask2=np_(mask2))
    # (iii) encoder
    xs = torch.randn(1, 25, 512, generator=g)
    ctx = torch.randn(1, 3, 512, generator=g)
    for name, c, streaming in (("enc_full", None, False), ("enc_ctx_stream", ctx, True)):
        kw = {} if c is None else {"context": c}
        h, m = flow.encoder(xs, torch.tensor([25]), streaming=streaming, **kw)
        out[name] = np_(h)
        print(name, tuple(h.shape), f"std {h.std():.3f}")
    out.update(enc_xs=np_(xs), enc_ctx=np_(ctx))
    # (iv) full flow.inference
    tok = torch.randint(0, 6561, (1, 25), generator=g)
    ptok = torch.randint(0, 6561, (1, 7), generator=g)
    pfeat = torch.randn(1, 14, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    none_tok, none_feat = torch.zeros(1, 0, dtype=torch.long), torch.zeros(1, 0, 80)
    cases = {
        "flow_noprompt": (tok, none_tok, none_feat, False, True),
        "flow_prompt": (tok, ptok, pfeat, False, True),
        "flow_stream_nofinal": (tok, ptok, pfeat, True, False),
        "flow_stream_final": (tok, ptok, pfeat, True, True),
    }
    for name, (tk, pt, pf, streaming, finalize) in cases.items():
        y, _ = flow.inference(token=tk, token_len=torch.tensor([tk.shape[1]]), prompt_token=pt,
                              prompt_token_len=torch.tensor([pt.shape[1]]), prompt_feat=pf,
                              prompt_feat_len=torch.tensor([pf.shape[1]]), embedding=emb, streaming=streaming,
                              finalize=finalize)
        out[name] = np_(y)
        print(name, tuple(y.shape), f"std {y.std():.3f} absmax {y.abs().max():.3f}")
    out.update(flow_tok=np_(tok), flow_ptok=np_(ptok), flow_pfeat=np_(pfeat), flow_emb=np_(emb))
    out["rand_noise_head"] = np_(flow.decoder.rand_noise[:, :, :64])
    np.savez_compressed(os.path.join(GOLD, "flow.npz"), **out)


def gen_blocks():
    """(ii, continued) every sub-block of the estimator on its own (SURVEY.md §8c "each sub-block"), run with the
    reference's classes: TimestepEmbedding, CausalBlock1D, CausalResnetBlock1D, FeedForward, BasicTransformerBlock (with a
    pad mask and with a chunk mask as additive bias), the Qwen2 backbone's forward_one_step on a 2-layer model, plus the
    non-causal matcha Block1D / ResnetBlock1D (GroupNorm) with small seeded weights that travel inside the fixture."""
    flow = build_flow()
    sd = flow.state_dict()
    flow.load_state_dict(W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED), strict=True)
    est = flow.decoder.estimator
    from cosyvoice.utils.mask import add_optional_chunk_mask
    from cosyvoice.utils.common import mask_to_bias
    out = {}
    g = torch.Generator().manual_seed(21)
    B, T = 2, 40
    with torch.no_grad():
        t = torch.tensor([0.25, 0.7])
        temb_in = est.time_embeddings(t)
        temb = est.time_mlp(temb_in)
        out.update(temb_in=np_(temb_in), temb=np_(temb))
        x = torch.randn(B, 256, T, generator=g)
        mask = torch.ones(B, 1, T)
        mask[1, :, 31:] = 0
        fb = est.final_block
        out.update(cb_x=np_(x), cb_mask=np_(mask), cb_out=np_(fb(x, mask)))
        rb = est.mid_blocks[0][0]
        out["rb_out"] = np_(rb(x, mask, temb))
        rb0 = est.down_blocks[0][0]
        x320 = torch.randn(B, 320, T, generator=g)
        out.update(rb0_x=np_(x320), rb0_out=np_(rb0(x320, mask, temb)))
        tb = est.mid_blocks[0][1][0]
        hs = torch.randn(B, T, 256, generator=g)
        out["tb_hs"] = np_(hs)
        for name, chunk in (("pad", 0), ("chunk", 16)):
            am = add_optional_chunk_mask(hs, mask.bool(), False, False, 0, chunk, -1).repeat(1, T, 1) if chunk == 0 else \
                add_optional_chunk_mask(hs, mask.bool(), False, False, 0, chunk, -1)
            bias = mask_to_bias(am == 1, hs.dtype)
            out[f"tb_bias_{name}"] = np_(bias)
            out[f"tb_out_{name}"] = np_(tb(hidden_states=hs, attention_mask=bias, timestep=None))
        out["ff_out"] = np_(tb.ff(hs))
        # non-causal matcha blocks (GroupNorm), small seeded weights
        from matcha.models.components.decoder import Block1D, ResnetBlock1D
        torch.manual_seed(5)
        mb, mr = Block1D(64, 96, groups=8).eval(), ResnetBlock1D(64, 96, 128, groups=8).eval()
        xs = torch.randn(B, 64, T, generator=g)
        te = torch.randn(B, 128, generator=g)
        out.update(m_x=np_(xs), m_te=np_(te), mb_out=np_(mb(xs, mask)), mr_out=np_(mr(xs, mask, te)))
        for k, v in mb.state_dict().items():
            out["mb." + k] = np_(v)
        for k, v in mr.state_dict().items():
            out["mr." + k] = np_(v)
    for k in ("temb", "cb_out", "rb_out", "rb0_out", "tb_out_pad", "tb_out_chunk", "ff_out", "mb_out", "mr_out"):
        print(k, out[k].shape, f"std {out[k].std():.3f}")
    np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **out)


def gen_llm():
    R.import_cosyvoice()
    from functools import partial
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM
    from cosyvoice.utils.common import ras_sampling
    # CosyVoice-BlankEN == Qwen2.5-0.5B shape; random init written to a temp dir
    cfg = Qwen2Config(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=24,
                      num_attention_heads=14, num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6,
                      tie_word_embeddings=True, max_position_embeddings=32768)
    d = tempfile.mkdtemp()
    t0 = time.time()
    Qwen2ForCausalLM(cfg).save_pretrained(d)
    enc = Qwen2Encoder(d)
    lm = Qwen2LM(896, 896, 6561, enc, partial(ras_sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1),
                 True, 0, [5, 15], use_speaker_encoder=False).eval()
    sd = {k: v for k, v in lm.state_dict().items() if k != "llm.model.lm_head.weight"}   # tied to embed_tokens
    W.save_manifest(sd, os.path.join(GOLD, "manifest_llm.json"))
    syn = W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED)
    syn["llm.model.lm_head.weight"] = syn["llm.model.model.embed_tokens.weight"]
    lm.load_state_dict(syn, strict=True)
    print("llm built", time.time() - t0)
    g = torch.Generator().manual_seed(21)
    text = torch.randint(0, 151936, (1, 12), generator=g)
    ptext = torch.randint(0, 151936, (1, 5), generator=g)
    pspeech = torch.randint(0, 6561, (1, 9), generator=g)
    forced = torch.randint(0, 6561, (16,), generator=g)
    # lm_input as Qwen2LM.inference builds it (llm.py:691-703)
    tk = torch.cat([ptext, text], dim=1)
    temb = lm.llm.model.model.embed_tokens(tk)
    sos = lm.llm_embedding.weight[0].reshape(1, 1, -1)
    task = lm.llm_embedding.weight[1].reshape(1, 1, -1)
    lm_input = torch.cat([sos, temb, task, lm.speech_embedding(pspeech)], dim=1)
    # full causal attention over the cache: drive HF with attention_mask=None
    logps, hiddens = [], []
    cache = None
    x = lm_input
    for i in range(17):
        o = lm.llm.model(inputs_embeds=x, output_hidden_states=True, return_dict=True, use_cache=True,
                         past_key_values=cache)
        cache = o.past_key_values
        y = o.hidden_states[-1]
        logp = lm.llm_decoder(y[:, -1]).log_softmax(dim=-1)
        logps.append(np_(logp[0]))
        if i == 0:
            hiddens = np_(y)
        if i < 16:
            x = lm.speech_embedding.weight[forced[i]].reshape(1, 1, -1)
    out = dict(text=np_(text), ptext=np_(ptext), pspeech=np_(pspeech), forced=np_(forced),
               lm_input=np_(lm_input), prefill_hidden=hiddens, logp=np.stack(logps))
    print("logp", out["logp"].shape, "max prob", float(np.exp(out["logp"]).max()))
    np.savez_compressed(os.path.join(GOLD, "llm.npz"), **out)


def gen_bistream():
    """Qwen2LM.inference_bistream (llm.py:762-870) of the reference on a 2-layer Qwen2 shape with a scripted
    `sampling` callable: the yielded tokens and the log-probs of every sampling call pin the interleaving logic."""
    R.import_cosyvoice()
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM
    from oracle.llm import ScriptedSampling
    cfg = Qwen2Config(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=2,
                      num_attention_heads=14, num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6,
                      tie_word_embeddings=True, max_position_embeddings=32768)
    d = tempfile.mkdtemp()
    Qwen2ForCausalLM(cfg).save_pretrained(d)
    out = {}
    cases = {
        # name: (prompt_text len, prompt_speech len, text chunk lens, fill_at, eos_from)
        "a": (5, 20, (3, 4, 6, 2), (12,), 40),     # prompt speech interleaved 15 + 5, sampled fill, forced fill later
        "b": (0, 0, (2, 2, 3, 5, 1), (15,), 38),   # no prompt: "not enough text, wait for more" branches
        "c": (2, 0, (2,), (), 12),                 # never 5 text tokens: everything happens in the final decode
    }
    for name, (npt, nps, chunks, fill_at, eos_from) in cases.items():
        rec = []
        samp = ScriptedSampling(fill_at, eos_from, record=rec)
        lm = Qwen2LM(896, 896, 6561, Qwen2Encoder(d), samp, True, 0, [5, 15], use_speaker_encoder=False).eval()
        sd = {k: v for k, v in lm.state_dict().items() if k != "llm.model.lm_head.weight"}
        syn = W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED)
        syn["llm.model.lm_head.weight"] = syn["llm.model.model.embed_tokens.weight"]
        lm.load_state_dict(syn, strict=True)
        g = torch.Generator().manual_seed(61 + len(out))
        ptext = torch.randint(0, 151936, (1, npt), generator=g)
        pspeech = torch.randint(0, 6561, (1, nps), generator=g)
        texts = [torch.randint(0, 151936, (1, n), generator=g) for n in chunks]
        toks = list(lm.inference_bistream(text=(t for t in texts), prompt_text=ptext, prompt_text_len=torch.tensor([npt]),
                                          prompt_speech_token=pspeech, prompt_speech_token_len=torch.tensor([nps]),
                                          embedding=torch.zeros(0, 192)))
        out[f"{name}_ptext"], out[f"{name}_pspeech"] = np_(ptext), np_(pspeech)
        out[f"{name}_text"] = np_(torch.cat(texts, dim=1))
        out[f"{name}_chunks"] = np.array(chunks)
        out[f"{name}_fill_at"], out[f"{name}_eos_from"] = np.array(fill_at, dtype=np.int64), np.array(eos_from)
        out[f"{name}_tokens"] = np.array([int(t) for t in toks])
        lp = torch.stack(rec)
        out[f"{name}_logp_head"] = np_(lp[:, :128])                       # leading ids + the maximum pin every call
        out[f"{name}_logp_max"] = np_(lp.max(dim=1).values)
        print(f"bistream {name}: {len(toks)} tokens yielded, {len(rec)} sampling calls")
    np.savez_compressed(os.path.join(GOLD, "bistream.npz"), **out)


def gen_spk():
    """LearnableSpeakerEncoder alone, flow.inference with reference_mels (use_speaker_encoder=True) and the lm_input of
    Qwen2LM.inference_spk."""
    R.import_cosyvoice()
    from cosyvoice.llm.llm import LearnableSpeakerEncoder
    torch.manual_seed(0)
    enc = LearnableSpeakerEncoder(mel_dim=80, model_dim=512, output_dim=192, num_blocks=6, num_heads=8).eval()
    sd = {"speaker_encoder." + k: v for k, v in enc.state_dict().items()}
    W.save_manifest(sd, os.path.join(GOLD, "manifest_spk.json"))
    syn = W.synth_state_dict({k: v.shape for k, v in sd.items()}, SEED)
    enc.load_state_dict({k[len("speaker_encoder."):]: v for k, v in syn.items()}, strict=True)
    g = torch.Generator().manual_seed(41)
    out = {}
    for T in (37, 150):
        mel = torch.randn(2, 80, T, generator=g)
        out[f"mel_T{T}"] = np_(mel)
        out[f"emb_T{T}"] = np_(enc(mel))
        print(f"spk T={T}: emb {tuple(out[f'emb_T{T}'].shape)} norm {np.linalg.norm(out[f'emb_T{T}'], axis=1)}")
    # flow with the speaker encoder enabled and two reference crops
    import cosyvoice.flow.flow as FL
    flow = build_flow_spk()
    fsd = flow.state_dict()
    W.save_manifest(fsd, os.path.join(GOLD, "manifest_flow_spk.json"))
    flow.load_state_dict(W.synth_state_dict({k: v.shape for k, v in fsd.items()}, SEED), strict=True)
    tok = torch.randint(0, 6561, (1, 20), generator=g)
    refs = torch.randn(1, 2, 80, 60, generator=g)
    y, _ = flow.inference(token=tok, token_len=torch.tensor([20]), prompt_token=torch.zeros(1, 0, dtype=torch.long),
                          prompt_token_len=torch.tensor([0]), prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=torch.tensor([0]),
                          embedding=None, reference_mels=refs, reference_mel_lengths=torch.tensor([[60, 60]]),
                          reference_mel_masks=torch.ones(1, 2, 60), streaming=False, finalize=True)
    out.update(flow_tok=np_(tok), flow_refs=np_(refs), flow_out=np_(y))
    print("flow with speaker encoder:", tuple(y.shape), float(y.std()))
    np.savez_compressed(os.path.join(GOLD, "spk.npz"), **out)


def build_flow_spk():
    from cosyvoice.flow.flow import CausalMaskedDiffWithXvec
    base = build_flow()
    return CausalMaskedDiffWithXvec(input_size=512, output_size=80, spk_embed_dim=192, output_type="mel", vocab_size=6561,
                                    input_frame_rate=25, only_mask_loss=True, token_latent_ratio=2, pre_lookahead_len=3,
                                    use_speaker_encoder=True, freeze_speaker_encoder=True, encoder=base.encoder,
                                    decoder=base.decoder).eval()


def gen_stream():
    """The hop schedule of the reference's own CosyVoice2Model.tts(stream=True) loop (cli/model.py:336-378), driven with stub
    llm / flow / hift objects: the llm stub yields N ids at once, the flow stub records (tokens seen, finalize, streaming) and
    returns zeros of the length flow.inference would (the prompt's frames dropped), the hift stub returns silence of the right
    length; token_offset is recovered from the length of the mel the hift stub is handed (:296 slices it off, :298-301 puts
    mel_cache_len cached frames in front).  Pins oracle/stream.py::hop_schedule to the reference."""
    R.import_cosyvoice()
    from cosyvoice.cli.model import CosyVoice2Model
    out = {}
    cases = [(118, 0), (130, 30), (20, 0), (28, 0), (1499, 75), (53, 25), (103, 7)]
    for N, Lp in cases:
        calls = []

        class Llm:
            def inference(self, **kw):
                for i in range(N):
                    yield i % 6561

        class Flow:
            pre_lookahead_len, token_mel_ratio, input_frame_rate = 3, 2, 25

            def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding,
                          streaming, finalize):
                calls.append([int(token.shape[1]), -1, int(bool(finalize)), int(bool(streaming))])
                return torch.zeros(1, 80, 2 * int(token.shape[1])), None

        class Hift:
            def inference(self, speech_feat, cache_source):
                rec = calls[-1]
                cached = 0 if cache_source.shape[2] == 0 else 8           # mel_cache_len frames of the previous pass in front
                rec[1] = rec[0] - (int(speech_feat.shape[2]) - cached) // 2
                n = int(speech_feat.shape[2]) * 480
                return torch.zeros(1, n), torch.zeros(1, 1, n)

        model = CosyVoice2Model(Llm(), Flow(), Hift())
        model.device = torch.device("cpu")
        chunks = list(model.tts(text=torch.zeros(1, 4, dtype=torch.int32), flow_embedding=torch.zeros(1, 192),
                                llm_embedding=torch.zeros(1, 192), flow_prompt_speech_token=torch.zeros(1, Lp, dtype=torch.int32),
                                prompt_speech_feat=torch.zeros(1, 2 * Lp, 80), stream=True))
        assert len(chunks) == len(calls)
        out[f"calls_{N}_{Lp}"] = np.array(calls, dtype=np.int64)
        print(f"stream N={N} Lp={Lp}: {len(calls)} flow calls; first {calls[0]}, last {calls[-1]}")
    out["cases"] = np.array(cases, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "stream.npz"), **out)


def gen_sampler():
    R.import_cosyvoice()
    from cosyvoice.utils.common import ras_sampling, nucleus_sampling, random_sampling
    ids, nuc, rnd = [], [], []
    for s, (logp, hist) in enumerate(W.sampler_cases()):
        torch.manual_seed(1000 + s)
        ids.append(int(ras_sampling(logp, hist, 25, top_p=0.8, top_k=25, win_size=10, tau_r=0.1)))
        torch.manual_seed(1000 + s)
        nuc.append(int(nucleus_sampling(logp, top_p=0.8, top_k=25)))
        torch.manual_seed(1000 + s)
        rnd.append(int(random_sampling(logp, hist, 25)))
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), ras=np.array(ids), nucleus=np.array(nuc),
                        random=np.array(rnd))
    print("sampler: 200 cases; distinct ids", len(set(ids)))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["dac", "dacfp32", "dacenc", "flow", "blocks", "llm", "bistream", "sampler", "spk", "stream"]
    for w in which:
        t0 = time.time()
        {"dac": gen_dac, "dacfp32": lambda: gen_dac("fp32"), "stream": gen_stream, "dacenc": gen_dac_enc, "flow": gen_flow, "blocks": gen_blocks, "llm": gen_llm, "bistream": gen_bistream, "sampler": gen_sampler, "spk": gen_spk}[w]()
        print(f"[{w}] done in {time.time() - t0:.1f}s")
