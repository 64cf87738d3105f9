"""oracle/spk.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU fp32 restatement of the learnable speaker encoder and of the two places it feeds.

Follows (reference, read-only; paths under speech/):
  cosyvoice/llm/llm.py:34-96              LearnableSpeakerEncoder (Conv1d k1 -> 6 AttentionBlocks -> frame 0 -> Linear -> L2 norm)
  cosyvoice/transformer/arch_util.py:21-38,41-77,80-123   GroupNorm32 (32 groups on x.float()), QKVAttentionLegacy, AttentionBlock
  cosyvoice/flow/flow.py:332-378,456-469  get_speaker_embedding (mean over references, normalise) -> spk_embed_affine_layer
  cosyvoice/llm/llm.py:163-188,616-674    get_speaker_conditioning / inference_spk ([sos | spk | text | task | prompt speech])
"""
import math

import torch
import torch.nn.functional as F


def attention_block(sd, p, x, heads=8):
    """AttentionBlock.forward on x [B, C, T] (mask and relative positions unused on this path)."""
    B, C, T = x.shape
    h = F.group_norm(x.float(), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    qkv = F.conv1d(h, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])                 # [B, 3C, T], heads are split FIRST
    ch = C // heads
    q, k, v = qkv.reshape(B * heads, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(B, C, T)
    return x + F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def speaker_encoder(sd, mel, prefix="speaker_encoder", blocks=6, heads=8):
    """mel [B, 80, T] -> L2-normalised [B, 192]."""
    h = F.conv1d(mel, sd[prefix + ".init.weight"], sd[prefix + ".init.bias"])
    for i in range(blocks):
        h = attention_block(sd, f"{prefix}.attn.{i}", h, heads)
    out = F.linear(h[:, :, 0], sd[prefix + ".output_proj.weight"], sd[prefix + ".output_proj.bias"])
    return F.normalize(out, p=2, dim=1)


def reference_embedding(sd, reference_mels, prefix="speaker_encoder"):
    """flow.py:338-368 / llm.py:166-184: [B, N, 80, T] (mean over the N references) or [B, 80, T] -> normalised [B, 192]."""
    if reference_mels.dim() == 4:
        e = torch.stack([speaker_encoder(sd, reference_mels[:, i], prefix) for i in range(reference_mels.shape[1])], 1).mean(1)
    else:
        e = speaker_encoder(sd, reference_mels, prefix)
    return F.normalize(e, dim=1)


def build_lm_input_spk(sd, text, prompt_text, prompt_speech_token, speaker_embed):
    """llm.py:634-665: [sos | speaker_embed | embed(prompt_text ++ text) | task_id | speech_emb(prompt)]."""
    tok = torch.cat([prompt_text, text], dim=1).long()
    t = F.embedding(tok, sd["llm.model.model.embed_tokens.weight"])
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    ps = (F.embedding(prompt_speech_token.long(), sd["speech_embedding.weight"]) if prompt_speech_token.shape[1]
          else torch.zeros(1, 0, t.shape[-1]))
    return torch.cat([sos, speaker_embed, t, task, ps], dim=1)
