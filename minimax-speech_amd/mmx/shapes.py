"""State-dict manifests (key -> shape) of the three hot-path models, generated from their constructor
arguments.  They are the single description of the reference's checkpoint layout in this package: the
drop-in nn.Module shells (mmx/shell.py) are built from them, so `state_dict()` / `load_state_dict()` use the
reference's key names, and tests/test_dropin_api.py checks them against the manifests captured
from the reference's own modules (tests/golden/manifest_*.json)."""
from typing import Dict, List, Tuple

Manifest = Dict[str, Tuple[int, ...]]


def dac_decoder_manifest(latent_dim=80, decoder_dim=1536, rates=(5, 4, 4, 3, 2), d_out=1) -> Manifest:
    """dac-vae/model.py: Decoder (:326-379) + de_conv_pre (:463); weight-norm g/v parametrisation."""
    m: Manifest = {}

    def wn(p, cout, cin, k):
        m[p + ".bias"] = (cout,)
        m[p + ".weight_g"] = (cout, 1, 1)
        m[p + ".weight_v"] = (cout, cin, k)

    wn("decoder.model.0.0", decoder_dim, latent_dim, 7)
    c = decoder_dim
    for i, s in enumerate(rates):
        q = f"decoder.model.{1 + i}.block"
        cin, cout = c, c // 2
        m[q + ".0.alpha"] = (1, cin, 1)
        # ConvTranspose1d weight is [Cin, Cout, 2s]; weight_g is per dim-0 slice (Cin); bias per Cout
        m[q + ".1.bias"] = (cout,)
        m[q + ".1.weight_g"] = (cin, 1, 1)
        m[q + ".1.weight_v"] = (cin, cout, 2 * s)
        for j in range(3):
            r = f"{q}.{2 + j}.block"
            m[r + ".0.alpha"] = (1, cout, 1)
            wn(r + ".1.0", cout, cout, 7)
            m[r + ".2.alpha"] = (1, cout, 1)
            wn(r + ".3.0", cout, cout, 1)
        c = cout
    n = len(rates)
    m[f"decoder.model.{n + 1}.alpha"] = (1, c, 1)
    wn(f"decoder.model.{n + 2}.0", d_out, c, 7)
    wn("de_conv_pre.0", latent_dim, latent_dim, 1)
    return m


def dac_encoder_manifest(latent_dim=80, encoder_dim=64, rates=(2, 3, 4, 4, 5), d_in=1) -> Manifest:
    """dac-vae/model.py: Encoder (:195-234) + EncoderBlock (:146-192) + en_conv_post (:459-461)."""
    m: Manifest = {}

    def wn(p, cout, cin, k):
        m[p + ".bias"] = (cout,)
        m[p + ".weight_g"] = (cout, 1, 1)
        m[p + ".weight_v"] = (cout, cin, k)

    wn("encoder.block.0.0", encoder_dim, d_in, 7)
    c = encoder_dim
    for i, s in enumerate(rates):
        q = f"encoder.block.{1 + i}.block"
        for j in range(3):
            r = f"{q}.{j}.block"
            m[r + ".0.alpha"] = (1, c, 1)
            wn(r + ".1.0", c, c, 7)
            m[r + ".2.alpha"] = (1, c, 1)
            wn(r + ".3.0", c, c, 1)
        m[q + ".3.alpha"] = (1, c, 1)
        wn(q + ".4.0", 2 * c, c, 2 * s)
        c *= 2
    n = len(rates)
    m[f"encoder.block.{n + 1}.alpha"] = (1, c, 1)
    wn(f"encoder.block.{n + 2}.0", latent_dim, c, 3)
    wn("en_conv_post.0", 2 * latent_dim, latent_dim, 1)
    return m


def speaker_encoder_manifest(prefix="speaker_encoder", mel_dim=80, model_dim=512, output_dim=192, num_blocks=6) -> Manifest:
    """LearnableSpeakerEncoder (speech/cosyvoice/llm/llm.py:34-63) with AttentionBlock (arch_util.py:87-115)."""
    m: Manifest = {f"{prefix}.init.weight": (model_dim, mel_dim, 1), f"{prefix}.init.bias": (model_dim,)}
    for i in range(num_blocks):
        p = f"{prefix}.attn.{i}"
        m[p + ".norm.weight"] = (model_dim,)
        m[p + ".norm.bias"] = (model_dim,)
        m[p + ".qkv.weight"] = (3 * model_dim, model_dim, 1)
        m[p + ".qkv.bias"] = (3 * model_dim,)
        m[p + ".proj_out.weight"] = (model_dim, model_dim, 1)
        m[p + ".proj_out.bias"] = (model_dim,)
    m[f"{prefix}.output_proj.weight"] = (output_dim, model_dim)
    m[f"{prefix}.output_proj.bias"] = (output_dim,)
    return m


def flow_manifest(vocab=6561, input_size=512, output_size=80, spk_embed_dim=192, heads=8, linear_units=2048,
                  num_blocks=6, num_up_blocks=4, est_in=320, est_ch=256, n_blocks=4, num_mid_blocks=12, est_heads=8,
                  head_dim=64, pre_lookahead_len=3, use_speaker_encoder=False) -> Manifest:
    """speech/config.yaml:60-116 -> CausalMaskedDiffWithXvec(UpsampleConformerEncoder, CausalConditionalCFM(
    CausalConditionalDecoder)) with use_speaker_encoder=False."""
    m: Manifest = {}
    d = input_size

    def lin(p, o, i, bias=True):
        m[p + ".weight"] = (o, i)
        if bias:
            m[p + ".bias"] = (o,)

    def ln(p, c):
        m[p + ".weight"] = (c,)
        m[p + ".bias"] = (c,)

    def conv(p, o, i, k):
        m[p + ".weight"] = (o, i, k)
        m[p + ".bias"] = (o,)

    m["input_embedding.weight"] = (vocab, d)
    if use_speaker_encoder:
        m.update(speaker_encoder_manifest(output_dim=spk_embed_dim))
    lin("spk_embed_affine_layer", output_size, spk_embed_dim)
    for e in ("encoder.embed", "encoder.up_embed"):
        lin(e + ".out.0", d, d)
        ln(e + ".out.1", d)
    conv("encoder.pre_lookahead_layer.conv1", d, d, pre_lookahead_len + 1)
    conv("encoder.pre_lookahead_layer.conv2", d, d, 3)
    for name, n in (("encoders", num_blocks), ("up_encoders", num_up_blocks)):
        for i in range(n):
            p = f"encoder.{name}.{i}"
            m[p + ".self_attn.pos_bias_u"] = (heads, d // heads)
            m[p + ".self_attn.pos_bias_v"] = (heads, d // heads)
            for q in ("linear_q", "linear_k", "linear_v", "linear_out"):
                lin(f"{p}.self_attn.{q}", d, d)
            lin(p + ".self_attn.linear_pos", d, d, bias=False)
            lin(p + ".feed_forward.w_1", linear_units, d)
            lin(p + ".feed_forward.w_2", d, linear_units)
            ln(p + ".norm_ff", d)
            ln(p + ".norm_mha", d)
    conv("encoder.up_layer.conv", d, d, 5)
    ln("encoder.after_norm", d)
    lin("encoder_proj", output_size, d)
    q = "decoder.estimator"
    C, inner, temb = est_ch, est_heads * head_dim, est_ch * 4
    lin(q + ".time_mlp.linear_1", temb, est_in)
    lin(q + ".time_mlp.linear_2", temb, temb)

    def resnet(p, cin):
        lin(p + ".mlp.1", C, temb)
        conv(p + ".block1.block.0", C, cin, 3)
        ln(p + ".block1.block.2", C)
        conv(p + ".block2.block.0", C, C, 3)
        ln(p + ".block2.block.2", C)
        conv(p + ".res_conv", C, cin, 1)

    def tblock(p):
        ln(p + ".norm1", C)
        for t in ("to_q", "to_k", "to_v"):
            lin(f"{p}.attn1.{t}", inner, C, bias=False)
        lin(p + ".attn1.to_out.0", C, inner)
        ln(p + ".norm3", C)
        lin(p + ".ff.net.0.proj", 4 * C, C)
        lin(p + ".ff.net.2", C, 4 * C)

    def stage(p, cin):
        resnet(p + ".0", cin)
        for j in range(n_blocks):
            tblock(f"{p}.1.{j}")

    stage(q + ".down_blocks.0", est_in)
    conv(q + ".down_blocks.0.2", C, C, 3)
    for i in range(num_mid_blocks):
        stage(f"{q}.mid_blocks.{i}", C)
    stage(q + ".up_blocks.0", 2 * C)
    conv(q + ".up_blocks.0.2", C, C, 3)
    conv(q + ".final_block.block.0", C, C, 3)
    ln(q + ".final_block.block.2", C)
    conv(q + ".final_proj", output_size, C, 1)
    return m


def llm_manifest(vocab=151936, hidden=896, inter=4864, layers=24, heads=14, kv_heads=2, head_dim=64,
                 speech_token_size=6561, spk_embed_dim=192, tie_lm_head=True, use_speaker_encoder=False) -> Manifest:
    """Qwen2LM (llm.py:375-436) around Qwen2Encoder(HF Qwen2ForCausalLM, CosyVoice-BlankEN == Qwen2.5-0.5B shape).
    `llm.model.lm_head.weight` is tied to embed_tokens and not listed (it is never read on the hot path)."""
    m: Manifest = {"llm_embedding.weight": (2, hidden), "llm.model.model.embed_tokens.weight": (vocab, hidden)}
    for l in range(layers):
        p = f"llm.model.model.layers.{l}"
        for n, o in (("q_proj", heads * head_dim), ("k_proj", kv_heads * head_dim), ("v_proj", kv_heads * head_dim)):
            m[f"{p}.self_attn.{n}.weight"] = (o, hidden)
            m[f"{p}.self_attn.{n}.bias"] = (o,)
        m[p + ".self_attn.o_proj.weight"] = (hidden, heads * head_dim)
        m[p + ".mlp.gate_proj.weight"] = (inter, hidden)
        m[p + ".mlp.up_proj.weight"] = (inter, hidden)
        m[p + ".mlp.down_proj.weight"] = (hidden, inter)
        m[p + ".input_layernorm.weight"] = (hidden,)
        m[p + ".post_attention_layernorm.weight"] = (hidden,)
    m["llm.model.model.norm.weight"] = (hidden,)
    m["llm_decoder.weight"] = (speech_token_size + 3, hidden)
    m["llm_decoder.bias"] = (speech_token_size + 3,)
    m["speech_embedding.weight"] = (speech_token_size + 3, hidden)
    if use_speaker_encoder:
        m.update(speaker_encoder_manifest(output_dim=spk_embed_dim))
    m["spk_embed_affine_layer.weight"] = (hidden, spk_embed_dim)
    m["spk_embed_affine_layer.bias"] = (hidden,)
    return m
