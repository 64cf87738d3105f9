"""oracle/llm.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU fp32 restatement of the autoregressive speech-token LM,
functional over the reference's state-dict keys (`llm.model.model.layers.N...`).

Follows (reference, read-only; paths under speech/):
  cosyvoice/llm/llm.py:676-711   Qwen2LM.inference  ([sos | text | task_id | prompt_speech] , min/max len)
  cosyvoice/llm/llm.py:745-760   inference_wrapper, non-vLLM branch (AR loop, stop/skip rules)
  cosyvoice/llm/llm.py:359-371   Qwen2Encoder.forward_one_step -> hidden_states[-1] of HF Qwen2ForCausalLM
  cosyvoice/llm/llm.py:259-274   sampling_ids (EOS re-draw loop, <=100 trials)
  cosyvoice/utils/common.py:111-139  ras_sampling / nucleus_sampling / random_sampling
Third party restated: HF transformers Qwen2 (pinned 4.40.1, requirements.txt:37; installed 5.15.0, source
  /usr/local/lib/python3.10/dist-packages/transformers/models/qwen2/modeling_qwen2.py: MLP :35-48, RoPE :51-135,
  eager attention :150-173 (softmax in fp32), RMSNorm :238-255 (fp32 variance), decoder layer :258-299).
Defined semantics (SURVEY.md §7 "version-drift trap"): every decode step attends causally over the
WHOLE KV cache (pinned-stack behaviour); the reference's (1,1) per-step mask is not replayed.

Sampling noise protocol: torch.multinomial(p, 1) on CPU is argmax(p / e) with e ~ Exp(1) drawn per
category from the default generator (verified against torch and against the reference's ras_sampling
in tests/test_oracle_sampling.py).  `*_e` variants take that noise explicitly ("injected noise"); with
oracle/philox.py as the source the GPU sampler consumes bit-identical draws.
"""
import math
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- Qwen2 backbone
class QwenCfg:
    def __init__(self, hidden=896, layers=24, heads=14, kv_heads=2, head_dim=64, inter=4864,
                 rope_theta=1e6, eps=1e-6, vocab=151936):
        self.hidden, self.layers, self.heads, self.kv_heads = hidden, layers, heads, kv_heads
        self.head_dim, self.inter, self.rope_theta, self.eps, self.vocab = head_dim, inter, rope_theta, eps, vocab


def rmsnorm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def rope_cos_sin(pos: torch.Tensor, dim: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    f = pos.float()[:, None] * inv[None, :]
    e = torch.cat([f, f], dim=-1)
    return e.cos(), e.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def qwen2_forward(sd, cfg: QwenCfg, x: torch.Tensor, cache: Optional[list], prefix="llm.model.model"):
    """x [1,q,H] appended after the `cache` (list of (k,v) per layer, [1,kvh,ctx,d]).
    Returns (final-RMSNorm hidden [1,q,H], new cache).  Full causal attention."""
    B, q, H = x.shape
    past = 0 if cache is None else cache[0][0].shape[2]
    pos = torch.arange(past, past + q)
    cos, sin = rope_cos_sin(pos, cfg.head_dim, cfg.rope_theta)
    g = cfg.heads // cfg.kv_heads
    new_cache = []
    h = x
    for l in range(cfg.layers):
        p = f"{prefix}.layers.{l}"
        r = rmsnorm(h, sd[p + ".input_layernorm.weight"], cfg.eps)
        qq = F.linear(r, sd[p + ".self_attn.q_proj.weight"], sd[p + ".self_attn.q_proj.bias"])
        kk = F.linear(r, sd[p + ".self_attn.k_proj.weight"], sd[p + ".self_attn.k_proj.bias"])
        vv = F.linear(r, sd[p + ".self_attn.v_proj.weight"], sd[p + ".self_attn.v_proj.bias"])
        qq = qq.view(B, q, cfg.heads, cfg.head_dim).transpose(1, 2)
        kk = kk.view(B, q, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        vv = vv.view(B, q, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        qq = qq * cos + rotate_half(qq) * sin
        kk = kk * cos + rotate_half(kk) * sin
        if cache is not None:
            kk = torch.cat([cache[l][0], kk], dim=2)
            vv = torch.cat([cache[l][1], vv], dim=2)
        new_cache.append((kk, vv))
        kr = kk.repeat_interleave(g, dim=1)
        vr = vv.repeat_interleave(g, dim=1)
        s = (qq @ kr.transpose(-2, -1)) * (cfg.head_dim ** -0.5)
        ctx = kk.shape[2]
        causal = torch.arange(ctx)[None, :] <= (past + torch.arange(q))[:, None]
        s = s.masked_fill(~causal, torch.finfo(s.dtype).min)
        a = torch.softmax(s, dim=-1, dtype=torch.float32).to(qq.dtype)
        o = (a @ vr).transpose(1, 2).reshape(B, q, cfg.heads * cfg.head_dim)
        h = h + F.linear(o, sd[p + ".self_attn.o_proj.weight"])
        r = rmsnorm(h, sd[p + ".post_attention_layernorm.weight"], cfg.eps)
        m = F.silu(F.linear(r, sd[p + ".mlp.gate_proj.weight"])) * F.linear(r, sd[p + ".mlp.up_proj.weight"])
        h = h + F.linear(m, sd[p + ".mlp.down_proj.weight"])
    return rmsnorm(h, sd[prefix + ".norm.weight"], cfg.eps), new_cache


# ----------------------------------------------------------------------------- sampling
def multinomial_e(p: torch.Tensor, e: torch.Tensor) -> int:
    """torch.multinomial(p, 1) on CPU == argmax(p / e) with e ~ Exp(1) drawn per category
    (ATen Distributions.cpp fast path for n_sample == 1); first maximum on ties."""
    return int(torch.argmax(p / e[: p.numel()]))


def nucleus_candidates(weighted_scores: torch.Tensor, top_p=0.8, top_k=25):
    """common.py:119-131: softmax, stable descending sort, keep while cum<top_p and n<top_k."""
    sv, si = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    prob, idx, cum = [], [], 0.0
    for i in range(len(si)):
        if cum < top_p and len(prob) < top_k:
            cum += sv[i]
            prob.append(sv[i])
            idx.append(si[i])
        else:
            break
    return torch.tensor(prob).to(weighted_scores), torch.tensor(idx, dtype=torch.long)


def ras_sampling_e(weighted_scores, decoded_tokens: List[int], noise: Callable[[int, int], torch.Tensor],
                   top_p=0.8, top_k=25, win_size=10, tau_r=0.1) -> int:
    """common.py:111-116 with injected noise: noise(which, n) -> e[n] (which 0 = nucleus, 1 = random)."""
    prob, idx = nucleus_candidates(weighted_scores, top_p, top_k)
    top = int(idx[multinomial_e(prob, noise(0, prob.numel()))])
    rep = sum(1 for t in decoded_tokens[-win_size:] if t == top)
    if rep >= win_size * tau_r:
        p = weighted_scores.softmax(dim=0)
        top = multinomial_e(p, noise(1, p.numel()))
    return top


def torch_noise(which: int, n: int) -> torch.Tensor:
    """The reference's own noise source: torch's default CPU generator, consumed in call order."""
    return torch.empty(n).exponential_(1)


def philox_noise(seed: int, seq: int, step: int, trial: int):
    from .philox import exp_noise
    return lambda which, n: torch.from_numpy(exp_noise(seed, seq, step, trial, which, n))


def sampling_ids_e(logp, decoded, noise_for_trial: Callable[[int], Callable], ignore_eos: bool, eos: int,
                   max_trials=100) -> int:
    """llm.py:259-274 with injected noise; trial k draws from noise_for_trial(k)."""
    trials = 0
    while True:
        top = ras_sampling_e(logp, decoded, noise_for_trial(trials))
        if (not ignore_eos) or top != eos:
            return top
        trials += 1
        if trials > max_trials:
            raise RuntimeError("sampling reaches max_trials {} and still get eos when ignore_eos is True".format(max_trials))


# ----------------------------------------------------------------------------- LM
def build_lm_input(sd, text, prompt_text, prompt_speech_token):
    """llm.py:691-703."""
    tok = torch.cat([prompt_text, text], dim=1).long()
    t = F.embedding(tok, sd["llm.model.model.embed_tokens.weight"])
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    if prompt_speech_token.shape[1] != 0:
        ps = F.embedding(prompt_speech_token.long(), sd["speech_embedding.weight"])
    else:
        ps = torch.zeros(1, 0, t.shape[-1], dtype=t.dtype)
    return torch.cat([sos, t, task, ps], dim=1)


def lm_inference(sd, cfg: QwenCfg, text, prompt_text, prompt_speech_token, seed: int = 0, seq: int = 0,
                 speech_token_size=6561, max_token_text_ratio=20, min_token_text_ratio=2,
                 forced: Optional[List[int]] = None, max_steps: Optional[int] = None,
                 ignore_eos_always=False, record: Optional[list] = None, unstable: Optional[list] = None,
                 sampled_out: Optional[list] = None) -> List[int]:
    """Qwen2LM.inference + inference_wrapper; sampling noise = philox(seed, seq, step, trial).
    `forced`: teacher forcing — step i feeds forced[i] as the accepted token (ids are still sampled
    and returned, so they can be compared step by step).  `record` collects per-step logp, `sampled_out` the id drawn at
    every step (ids above EOS included), `unstable` the steps whose draw a 6e-5 log-prob perturbation can flip
    (decision_unstable)."""
    lm_input = build_lm_input(sd, text, prompt_text, prompt_speech_token)
    tl = text.shape[1]
    min_len, max_len = int(tl * min_token_text_ratio), int(tl * max_token_text_ratio)
    if max_steps is not None:
        max_len = min(max_len, max_steps)
    out, sampled, cache = [], [], None
    for i in range(max_len):
        y, cache = qwen2_forward(sd, cfg, lm_input, cache)
        logp = F.linear(y[:, -1], sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(dim=-1).squeeze(0)
        if record is not None:
            record.append(logp.clone())
        top = sampling_ids_e(logp, out, lambda k, i=i: philox_noise(seed, seq, i, k),
                             ignore_eos=(ignore_eos_always or i < min_len), eos=speech_token_size)
        if unstable is not None and decision_unstable(logp, out, lambda k, i=i: philox_noise(seed, seq, i, k),
                                                      ignore_eos_always or i < min_len, speech_token_size):
            unstable.append(i)
        sampled.append(top)
        if sampled_out is not None:
            sampled_out.append(top)
        if forced is not None:
            top = forced[i]
        if top == speech_token_size:
            break
        if top > speech_token_size:
            continue          # llm.py:755-756: no yield and lm_input is NOT updated
        out.append(top)
        lm_input = sd["speech_embedding.weight"][top].reshape(1, 1, -1)
    return sampled if forced is not None else out


def decision_unstable(logp: torch.Tensor, decoded: List[int], noise_for_trial: Callable[[int], Callable], ignore_eos: bool, eos: int,
                      tol: float = 6e-5, top_p=0.8, top_k=25) -> bool:
    """Is the reference's draw at this step decided by a near-tie that a log-prob perturbation of `tol` can flip?  (Any two
    fp32 implementations of the LM differ by ~1e-5 in log-prob - the reference on another BLAS included - so over a long
    utterance such steps are where token ids can legitimately part.)  The reference's sampler (common.py:111-139) has three
    kinds of decision: the ORDER of the stable descending sort (the multinomial noise e_i goes by sorted position, so
    swapping two nearly equal candidates hands each the other's noise), the nucleus CUT (cum < top_p), and the RACES
    argmax p_i / e_i (nucleus and, on repetition, full vocabulary).  Unstable = the sampled id changes when two adjacent
    sorted candidates closer than tol swap their values, or the running sum passes top_p within tol, or a race's two
    largest ratios are within 2 tol (relative)."""
    base = sampling_ids_e(logp, decoded, noise_for_trial, ignore_eos, eos)
    sv, si = logp.sort(descending=True, stable=True)
    p = logp.softmax(0)
    prob, idx = nucleus_candidates(logp, top_p, top_k)
    n = prob.numel()
    cum = torch.cumsum(p[si[:n + 1]], 0)
    if min(abs(float(cum[n - 1]) - top_p), abs(float(cum[n - 2]) - top_p) if n > 1 else 1.0) < tol and n < top_k:
        return True
    for i in range(min(n, logp.numel() - 1)):            # adjacent pairs (i, i + 1) of the sorted prefix, boundary element included
        if float(sv[i] - sv[i + 1]) < tol:
            q = logp.clone()
            q[si[i]], q[si[i + 1]] = logp[si[i + 1]], logp[si[i]]
            if sampling_ids_e(q, decoded, noise_for_trial, ignore_eos, eos) != base:
                return True
    trial = 0
    while True:                                           # the races of every trial the draw went through
        noise = noise_for_trial(trial)
        r = (prob / noise(0, n)).sort(descending=True)
        if n > 1 and float((r.values[0] - r.values[1]) / r.values[0]) < 2 * tol:
            return True
        top = int(idx[r.indices[0]])
        if sum(1 for t in decoded[-10:] if t == top) >= 1:
            rr = (p / noise(1, p.numel())).sort(descending=True)
            if float((rr.values[0] - rr.values[1]) / rr.values[0]) < 2 * tol:
                return True
            top = int(rr.indices[0])
        if not ignore_eos or top != eos:
            return False
        trial += 1
        if trial > 100:
            return True


def lm_inference_bistream(sd, cfg: QwenCfg, text_chunks, prompt_text, prompt_speech_token,
                          sampling_ids: Callable[[torch.Tensor, List[int], int, bool], int],
                          mix_ratio=(5, 15), speech_token_size=6561, record: Optional[list] = None,
                          max_calls: int = 4096):
    """Qwen2LM.inference_bistream (speech/cosyvoice/llm/llm.py:762-870): text arrives in chunks and is interleaved
    with speech tokens mix_ratio[0] : mix_ratio[1]; the model asks for more text with the fill token
    (speech_token_size + 2), which is also forced every mix_ratio[1]+1 tokens once it has been seen.

    `sampling_ids(logp, out_tokens, call_index, ignore_eos)` stands for self.sampling_ids (llm.py:259-274);
    call_index counts LM forward passes (the forced-fill passes included, llm.py:824-827).
    Returns (yielded tokens, out_tokens incl. fill / eos).  `record` collects the logp of every pass."""
    fill, eos = speech_token_size + 2, speech_token_size
    emb = lambda ids: F.embedding(ids, sd["llm.model.model.embed_tokens.weight"])
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    H = sos.shape[-1]
    pse = F.embedding(prompt_speech_token, sd["speech_embedding.weight"]) if prompt_speech_token.shape[1] else torch.zeros(1, 0, H)
    lm_input = sos
    out_tokens, yielded, cache = [], [], None
    text_cache = emb(prompt_text)
    next_fill_index = -1
    calls = 0

    def forward(x):
        nonlocal cache, calls
        y, cache = qwen2_forward(sd, cfg, x, cache)
        logp = F.linear(y[:, -1], sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(dim=-1).squeeze(0)
        if record is not None:
            record.append(logp.clone())
        calls += 1
        if calls > max_calls:
            raise RuntimeError("bistream: no fill token / eos within max_calls")
        return logp

    for this_text in text_chunks:
        text_cache = torch.cat([text_cache, emb(this_text)], dim=1)
        while pse.shape[1] != 0:                                       # llm.py:796-805
            if text_cache.shape[1] >= mix_ratio[0]:
                lm_input = torch.cat([lm_input, text_cache[:, :mix_ratio[0]], pse[:, :mix_ratio[1]]], dim=1)
                text_cache, pse = text_cache[:, mix_ratio[0]:], pse[:, mix_ratio[1]:]
            else:
                break
        if pse.shape[1] == 0:                                          # llm.py:807-845
            if (len(out_tokens) != 0 and out_tokens[-1] == fill) or (len(out_tokens) == 0 and lm_input.shape[1] == 1):
                if text_cache.shape[1] >= mix_ratio[0]:
                    lm_input_text = text_cache[:, :mix_ratio[0]]
                    if len(out_tokens) != 0 and out_tokens[-1] == fill:
                        lm_input = lm_input_text
                    else:
                        lm_input = torch.cat([lm_input, lm_input_text], dim=1)
                    text_cache = text_cache[:, mix_ratio[0]:]
                else:
                    continue
            while True:
                logp = forward(lm_input)
                if next_fill_index != -1 and len(out_tokens) == next_fill_index:
                    top = fill
                    next_fill_index += mix_ratio[1] + 1
                else:
                    top = sampling_ids(logp, out_tokens, calls - 1, True)
                if top == fill:
                    next_fill_index = len(out_tokens) + mix_ratio[1] + 1
                out_tokens.append(top)
                if top >= speech_token_size:
                    if top == fill:
                        break
                    raise ValueError(f"should not get token {top}")
                yielded.append(top)
                lm_input = sd["speech_embedding.weight"][top].reshape(1, 1, -1)
    lm_input = torch.cat([lm_input, text_cache, task], dim=1)           # llm.py:848-870
    while True:
        logp = forward(lm_input)
        top = sampling_ids(logp, out_tokens, calls - 1, False)
        out_tokens.append(top)
        if top >= speech_token_size:
            if top == eos:
                break
            raise ValueError(f"should not get token {top}")
        yielded.append(top)
        lm_input = sd["speech_embedding.weight"][top].reshape(1, 1, -1)
    return yielded, out_tokens


class ScriptedSampling:
    """A deterministic `sampling` callable (the constructor argument of Qwen2LM, llm.py:381) used to pin the bistream
    control flow against the reference: argmax over the speech ids, the fill token when len(decoded) is in `fill_at`,
    eos once len(decoded) >= eos_from (and an argmax on the immediate retry that ignore_eos=True provokes)."""

    def __init__(self, fill_at=(), eos_from=10 ** 9, speech_token_size=6561, record=None):
        self.fill_at, self.eos_from, self.n = set(fill_at), eos_from, speech_token_size
        self.record = record
        self._last_eos_len = -1

    def __call__(self, weighted_scores, decoded_tokens, sampling):
        n = len(decoded_tokens)
        if n > 1000:
            raise RuntimeError("ScriptedSampling: runaway decode (no fill token / eos scheduled)")
        if self.record is not None and self._last_eos_len != n:
            self.record.append(weighted_scores.detach().float().cpu().clone())
        if n in self.fill_at:
            return torch.tensor(self.n + 2)
        if n >= self.eos_from and self._last_eos_len != n:
            self._last_eos_len = n
            return torch.tensor(self.n)
        return weighted_scores[:self.n].argmax()
