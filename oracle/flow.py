"""oracle/flow.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU fp32 restatement of the CosyVoice2 causal flow-matching decoder (SURVEY.md §8a rows
a6-a9), functional over the reference's state-dict keys.  Every tensor is [B, T, C]
("time-major rows") unless noted; the reference alternates [B,C,T] / [B,T,C].

Follows (reference, read-only; paths under speech/):
  cosyvoice/flow/flow.py:437-511              CausalMaskedDiffWithXvec.inference
  cosyvoice/transformer/upsample_encoder.py:243-316   UpsampleConformerEncoder.forward
      :66-102 PreLookaheadLayer, :37-63 Upsample1D
  cosyvoice/transformer/subsampling.py:69-113  LinearNoSubsampling (Linear+LayerNorm eps 1e-5)
  cosyvoice/transformer/embedding.py:201-302   EspnetRelPositionalEncoding (x*sqrt(d), pe table)
  cosyvoice/transformer/encoder_layer.py:160-236  ConformerEncoderLayer (pre-norm, no macaron/cnn, eps 1e-12)
  cosyvoice/transformer/attention.py:225-330   RelPositionMultiHeadedAttention (+rel_shift)
  cosyvoice/transformer/positionwise_feed_forward.py:47-55  FFN with SiLU ("swish")
  cosyvoice/flow/flow_matching.py:323-348,74-126   CausalConditionalCFM.forward / solve_euler
  cosyvoice/flow/decoder.py:405-496            CausalConditionalDecoder.forward (the estimator)
      :36-85 CausalConv1d / CausalBlock1D / CausalResnetBlock1D
  matcha/models/components/decoder.py:14-29,56-61,73-117  SinusoidalPosEmb, ResnetBlock1D.forward, TimestepEmbedding
  matcha/models/components/transformer.py:243-316  BasicTransformerBlock.forward
  cosyvoice/utils/mask.py:127-158,161-236,239-265   chunk / pad masks
  cosyvoice/utils/common.py:160-168            mask_to_bias: (1-m) * -1e10
Third party restated (absent offline): diffusers==0.29.0 Attention (AttnProcessor2_0: softmax(q k^T/sqrt(d)
  + bias) v, to_q/k/v no bias, to_out bias) and GELU (Linear + exact-erf gelu).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- masks
def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    max_len = max_len if max_len > 0 else int(lengths.max())
    return torch.arange(max_len)[None, :] >= lengths[:, None].to(torch.int64)


def subsequent_chunk_mask(size: int, chunk: int) -> torch.Tensor:
    # mask.py:154-157 (num_left_chunks is ignored by the reference)
    pos = torch.arange(size)
    return pos[None, :] < ((pos // chunk + 1) * chunk)[:, None]


def chunk_mask(masks: torch.Tensor, size: int, static_chunk: int) -> torch.Tensor:
    """add_optional_chunk_mask(xs, masks, False, False, 0, static_chunk, -1): masks [B,1,T] bool."""
    if static_chunk > 0:
        cm = masks & subsequent_chunk_mask(size, static_chunk)[None]
    else:
        cm = masks
    cm = cm.clone()
    dead = cm.sum(dim=-1) == 0          # mask.py:233-235: all-false rows are forced true
    cm[dead] = True
    return cm


# ----------------------------------------------------------------------------- encoder
def espnet_rel_pe(T: int, d: int) -> torch.Tensor:
    """position_encoding(offset=0, size=T) of EspnetRelPositionalEncoding: rows are relative
    positions T-1, ..., 0, ..., -(T-1)  -> [2T-1, d]  (embedding.py:233-253,296-302)."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def espnet_rel_pe_offset(T: int, d: int, offset: int) -> torch.Tensor:
    """embedding.py:296-302 with offset>0 (used for the look-ahead `context` embed): slice
    [c - T - offset + 1 : c + T + offset] of the table, i.e. relative positions
    T+offset-1 ... -(T+offset-1)."""
    return espnet_rel_pe(T + offset, d)


def rel_shift(x: torch.Tensor) -> torch.Tensor:
    """attention.py:225-247 in closed form: out[..., i, j] = x[..., i, T-1-i+j], j < T."""
    T = x.shape[-2]
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    return torch.gather(x, -1, (T - 1 - i + j).expand(*x.shape[:-2], T, T))


def rel_mha(sd, p, x, mask, pos_emb, heads=8):
    """RelPositionMultiHeadedAttention.forward on query=key=value=x [B,T,C]; mask [B,1|T,T] bool."""
    B, T, C = x.shape
    dk = C // heads
    q = F.linear(x, sd[p + ".linear_q.weight"], sd[p + ".linear_q.bias"]).view(B, T, heads, dk)
    k = F.linear(x, sd[p + ".linear_k.weight"], sd[p + ".linear_k.bias"]).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(x, sd[p + ".linear_v.weight"], sd[p + ".linear_v.bias"]).view(B, T, heads, dk).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + ".linear_pos.weight"]).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + ".pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[p + ".pos_bias_v"]).transpose(1, 2)
    ac = qu @ k.transpose(-2, -1)
    bd = qv @ pp.transpose(-2, -1)
    if ac.shape != bd.shape:
        bd = rel_shift(bd)
    scores = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    scores = scores.masked_fill(m, -float("inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = (attn @ v).transpose(1, 2).reshape(B, T, C)
    return F.linear(o, sd[p + ".linear_out.weight"], sd[p + ".linear_out.bias"])


def conformer_layer(sd, p, x, mask, pos_emb):
    # encoder_layer.py:198-236 with macaron=None, conv_module=None, normalize_before=True
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm_mha.weight"], sd[p + ".norm_mha.bias"], 1e-12)
    x = x + rel_mha(sd, p + ".self_attn", h, mask, pos_emb)
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm_ff.weight"], sd[p + ".norm_ff.bias"], 1e-12)
    h = F.linear(F.silu(F.linear(h, sd[p + ".feed_forward.w_1.weight"], sd[p + ".feed_forward.w_1.bias"])),
                 sd[p + ".feed_forward.w_2.weight"], sd[p + ".feed_forward.w_2.bias"])
    return x + h


def linear_embed(sd, p, x, offset=0):
    """LinearNoSubsampling + EspnetRelPositionalEncoding: returns (x*sqrt(d), pos_emb[1,2T'-1,d])."""
    d = sd[p + ".out.0.weight"].shape[0]
    x = F.linear(x, sd[p + ".out.0.weight"], sd[p + ".out.0.bias"])
    x = F.layer_norm(x, (d,), sd[p + ".out.1.weight"], sd[p + ".out.1.bias"], 1e-5)
    return x * math.sqrt(d), espnet_rel_pe_offset(x.shape[1], d, offset)[None]


def pre_lookahead(sd, p, x, context=None, L=3):
    # upsample_encoder.py:83-102
    o = x.transpose(1, 2)
    if context is None or context.shape[1] == 0:
        o = F.pad(o, (0, L))
    else:
        assert context.shape[1] == L
        o = torch.cat([o, context.transpose(1, 2)], dim=2)
    o = F.leaky_relu(F.conv1d(o, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]))   # slope 0.01
    o = F.pad(o, (2, 0))
    o = F.conv1d(o, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])
    return o.transpose(1, 2) + x


def encoder_forward(sd, p, xs, xs_lens, context=None, streaming=False, static_chunk_size=25,
                    n_blocks=6, n_up_blocks=4, return_stages=False):
    """UpsampleConformerEncoder.forward: xs [B,T,512] -> ([B,2T,512], masks [B,1,2T])."""
    stages = {}
    T = xs.shape[1]
    masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
    xs, pos_emb = linear_embed(sd, p + ".embed", xs)
    if context is not None and context.shape[1] != 0:
        context, _ = linear_embed(sd, p + ".embed", context, offset=xs.shape[1])
    cm = chunk_mask(masks, T, static_chunk_size if streaming else 0)
    xs = pre_lookahead(sd, p + ".pre_lookahead_layer", xs, context)
    stages["pre"] = xs
    for i in range(n_blocks):
        xs = conformer_layer(sd, f"{p}.encoders.{i}", xs, cm, pos_emb)
    stages["enc"] = xs
    # Upsample1D: nearest x2, left pad 4, conv k5 (upsample_encoder.py:59-63)
    o = F.interpolate(xs.transpose(1, 2), scale_factor=2.0, mode="nearest")
    o = F.pad(o, (4, 0))
    xs = F.conv1d(o, sd[p + ".up_layer.conv.weight"], sd[p + ".up_layer.conv.bias"]).transpose(1, 2)
    xs_lens = xs_lens * 2
    T = xs.shape[1]
    masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
    xs, pos_emb = linear_embed(sd, p + ".up_embed", xs)
    stages["up"] = xs
    cm = chunk_mask(masks, T, static_chunk_size * 2 if streaming else 0)
    for i in range(n_up_blocks):
        xs = conformer_layer(sd, f"{p}.up_encoders.{i}", xs, cm, pos_emb)
    xs = F.layer_norm(xs, xs.shape[-1:], sd[p + ".after_norm.weight"], sd[p + ".after_norm.bias"], 1e-5)
    return (xs, masks, stages) if return_stages else (xs, masks)


# ----------------------------------------------------------------------------- estimator
def sinusoidal_pos_emb(t: torch.Tensor, dim: int, scale: float = 1000.0) -> torch.Tensor:
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = scale * t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def causal_block(sd, p, x, mask):
    """CausalBlock1D: (x*mask) -> causal conv k3 -> LayerNorm(C) -> Mish -> *mask. x [B,T,Cin], mask [B,T,1]."""
    o = F.pad((x * mask).transpose(1, 2), (2, 0))
    o = F.conv1d(o, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"]).transpose(1, 2)
    o = F.layer_norm(o, o.shape[-1:], sd[p + ".block.2.weight"], sd[p + ".block.2.bias"], 1e-5)
    return F.mish(o) * mask


def causal_resnet(sd, p, x, mask, temb):
    h = causal_block(sd, p + ".block1", x, mask)
    h = h + F.linear(F.mish(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, None, :]
    h = causal_block(sd, p + ".block2", h, mask)
    res = F.conv1d((x * mask).transpose(1, 2), sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"]).transpose(1, 2)
    return h + res


def basic_transformer_block(sd, p, x, bias, heads=8, dh=64):
    """x [B,T,C]; bias [B,T,T] additive float (0 / -1e10)."""
    B, T, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    q = F.linear(h, sd[p + ".attn1.to_q.weight"]).view(B, T, heads, dh).transpose(1, 2)
    k = F.linear(h, sd[p + ".attn1.to_k.weight"]).view(B, T, heads, dh).transpose(1, 2)
    v = F.linear(h, sd[p + ".attn1.to_v.weight"]).view(B, T, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) * (dh ** -0.5) + bias[:, None]
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, heads * dh)
    x = x + F.linear(o, sd[p + ".attn1.to_out.0.weight"], sd[p + ".attn1.to_out.0.bias"])
    h = F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"]))
    return x + F.linear(h, sd[p + ".ff.net.2.weight"], sd[p + ".ff.net.2.bias"])


def estimator_forward(sd, p, x, mask, mu, t, spks, cond, streaming=False, static_chunk_size=50,
                      n_blocks=4, n_mid=None, heads=8, dh=64, return_stages=False):
    """CausalConditionalDecoder.forward with channels=[256] (config.yaml:105-116).
    Reference layout in/out: x, mu, cond [B,80,T]; mask [B,1,T]; t [B]; spks [B,80] -> [B,80,T]."""
    stages = {}
    B, _, T = x.shape
    if n_mid is None:                                  # config.yaml:111 num_mid_blocks = 12; reduced-depth tests carry fewer
        pre = p + ".mid_blocks."
        n_mid = 1 + max(int(k[len(pre):].split(".", 1)[0]) for k in sd if k.startswith(pre))
    temb = sinusoidal_pos_emb(t, sd[p + ".time_mlp.linear_1.weight"].shape[1]).to(t.dtype)
    temb = F.linear(F.silu(F.linear(temb, sd[p + ".time_mlp.linear_1.weight"], sd[p + ".time_mlp.linear_1.bias"])),
                    sd[p + ".time_mlp.linear_2.weight"], sd[p + ".time_mlp.linear_2.bias"])
    stages["temb"] = temb
    h = torch.cat([x, mu, spks[:, :, None].expand(-1, -1, T), cond], dim=1).transpose(1, 2)   # [B,T,320]
    m = mask.transpose(1, 2)                                                                    # [B,T,1]
    mb = mask.bool()
    if streaming:
        am = chunk_mask(mb, T, static_chunk_size)
    else:
        am = chunk_mask(mb, T, 0).repeat(1, T, 1)
    bias = (1.0 - am.to(x.dtype)) * -1.0e10

    def stage(q, h):
        h = causal_resnet(sd, q + ".0", h, m, temb)
        for j in range(n_blocks):
            h = basic_transformer_block(sd, f"{q}.1.{j}", h, bias, heads, dh)
        return h

    h = stage(p + ".down_blocks.0", h)
    stages["down"] = h
    skip = h
    # "downsample" of the last (only) down block is a CausalConv1d k3 (decoder.py:347-349)
    o = F.pad((h * m).transpose(1, 2), (2, 0))
    h = F.conv1d(o, sd[p + ".down_blocks.0.2.weight"], sd[p + ".down_blocks.0.2.bias"]).transpose(1, 2)
    for i in range(n_mid):
        h = stage(f"{p}.mid_blocks.{i}", h)
    stages["mid"] = h
    h = torch.cat([h, skip], dim=-1)
    h = stage(p + ".up_blocks.0", h)
    o = F.pad((h * m).transpose(1, 2), (2, 0))
    h = F.conv1d(o, sd[p + ".up_blocks.0.2.weight"], sd[p + ".up_blocks.0.2.bias"]).transpose(1, 2)
    stages["up"] = h
    h = causal_block(sd, p + ".final_block", h, m)
    o = F.conv1d((h * m).transpose(1, 2), sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"])
    o = o * mask
    return (o, stages) if return_stages else o


# ----------------------------------------------------------------------------- CFM / flow
def rand_noise(T: int = 15000) -> torch.Tensor:
    """CausalConditionalCFM.__init__ (flow_matching.py:320-321): torch CPU manual_seed(0); randn([1,80,15000])."""
    g = torch.Generator().manual_seed(0)
    return torch.randn([1, 80, T], generator=g)


def cosine_t_span(n: int) -> torch.Tensor:
    t = torch.linspace(0, 1, n + 1)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def cfm_forward(sd, p, mu, mask, spks, cond, n_timesteps=10, streaming=False, cfg=0.7, noise=None,
                return_steps=False):
    """CausalConditionalCFM.forward + solve_euler: mu, cond [1,80,T]; mask [1,1,T]; spks [1,80]."""
    T = mu.shape[2]
    z = (rand_noise() if noise is None else noise)[:, :, :T].to(mu.dtype)
    t_span = cosine_t_span(n_timesteps).to(mu.dtype)
    x = z
    t, dt = t_span[0:1], t_span[1] - t_span[0]
    steps = []
    for step in range(1, n_timesteps + 1):
        x_in = torch.cat([x, x], 0)
        mask_in = torch.cat([mask, mask], 0).to(x.dtype)
        mu_in = torch.cat([mu, torch.zeros_like(mu)], 0)
        t_in = torch.cat([t, t], 0)
        spks_in = torch.cat([spks, torch.zeros_like(spks)], 0)
        cond_in = torch.cat([cond, torch.zeros_like(cond)], 0)
        d = estimator_forward(sd, p + ".estimator", x_in, mask_in, mu_in, t_in, spks_in, cond_in, streaming)
        d = (1.0 + cfg) * d[0:1] - cfg * d[1:2]
        x = x + dt * d
        t = t + dt
        steps.append(x)
        if step < n_timesteps:
            dt = t_span[step + 1] - t
    return (x.float(), steps) if return_steps else x.float()


def flow_inference(sd, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True,
                   pre_lookahead_len=3, n_timesteps=10, return_parts=False):
    """CausalMaskedDiffWithXvec.inference (flow.py:437-511), batch 1.
    token [1,Lt] int, prompt_token [1,Lp] int, prompt_feat [1,Tp,80], embedding [1,192] -> [1,80,T2]."""
    assert token.shape[0] == 1
    emb = F.normalize(embedding, dim=1)
    emb = F.linear(emb, sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
    tok = torch.cat([prompt_token, token], dim=1).long()
    tok_len = torch.tensor([tok.shape[1]])
    x = F.embedding(torch.clamp(tok, min=0), sd["input_embedding.weight"])
    if finalize:
        h, _ = encoder_forward(sd, "encoder", x, tok_len, None, streaming)
    else:
        x, ctx = x[:, :-pre_lookahead_len], x[:, -pre_lookahead_len:]
        # NB flow.py:487-489 passes the un-shortened token_len; with B=1 the pad mask built from it
        # over T = Lt-3 columns is all-true, identical to using the shortened length.
        h, _ = encoder_forward(sd, "encoder", x, torch.tensor([x.shape[1]]), ctx, streaming)
    mel_len1 = prompt_feat.shape[1]
    mel_len2 = h.shape[1] - mel_len1
    mu = F.linear(h, sd["encoder_proj.weight"], sd["encoder_proj.bias"]).transpose(1, 2).contiguous()
    cond = torch.zeros(1, mel_len1 + mel_len2, 80)
    cond[:, :mel_len1] = prompt_feat
    cond = cond.transpose(1, 2)
    mask = torch.ones(1, 1, mel_len1 + mel_len2)
    feat = cfm_forward(sd, "decoder", mu, mask, emb, cond, n_timesteps, streaming)
    out = feat[:, :, mel_len1:].float()
    return (out, {"mu": mu, "spks": emb}) if return_parts else out
