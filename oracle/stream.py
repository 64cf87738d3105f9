"""oracle/stream.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the reference's streaming synthesis schedule, CosyVoice2Model.tts(stream=True)
(/root/reference/speech/cosyvoice/cli/model.py:321-386 with token2wav :285-319):

  * hop loop (:336-366): prompt_token_pad = ceil(Lp / 25) * 25 - Lp (:338); the first hop takes 25 + pad tokens, the later
    ones 25 (:341); a hop runs as soon as `hop + pre_lookahead_len` tokens beyond `token_offset` exist (:342), on the tokens
    [0, token_offset + hop + pre_lookahead_len) (:343) with streaming=True, finalize=False (:344-351);
  * token_offset slicing (:296): of a call's latents only the frames from token_offset * token_mel_ratio on are new;
  * closing pass (:369-378): ALL tokens, finalize=True and — the reference leaves `stream` at its default False here —
    WITHOUT the chunk masks, so it re-solves every frame a little differently from the streaming passes.

Every flow call is oracle.flow.flow_inference (pinned by the reference's flow_stream_* goldens, tests/test_oracle_golden.py).

The vocoder side differs by construction (BASELINE.json: the path renders with the DAC-VAE decoder, not with HiFT, whose
mel / source caches cli/model.py:298-311 manage).  The DAC decoder is a finite-context convolution stack, so the rule —
stated in mmx/pipeline.py::tts_stream, restated here — is:

  * a pass renders the frames whose right context (ctx_right latent frames) is final in that pass; the window it decodes
    starts ctx_left frames earlier, and those context frames are the values ALREADY RENDERED (streaming passes agree on
    finished frames, the closing pass does not);
  * like the reference (:306-311: every non-final pass keeps its last source_cache_len samples back), a streaming pass
    holds the samples of its last MEL_CACHE = 8 rendered frames back.  If the next pass is another streaming pass the
    held samples are emitted unchanged (the two passes agree, a cross-fade of equal signals would only add the window's
    gain: the halves of np.hamming(2n) sum to 1.08, not 1); if it is the CLOSING pass — the one seam where two passes
    disagree — the closing pass renders those 8 frames again from its own latents and the two renderings are cross-faded
    with fade_in_out (utils/common.py:142-150) over the window of cli/model.py:262, np.hamming(2 * 8 * hop), normalised
    so that its two halves sum to one.
"""
import math
from typing import List, Tuple

import numpy as np
import torch

from . import dac as ODAC
from . import flow as OFLOW

TOKEN_HOP = 25            # cli/model.py:256
MEL_CACHE = 8             # cli/model.py:258: frames of overlap between two passes


def hop_schedule(n_tokens: int, prompt_len: int, lookahead: int = 3, token_hop: int = TOKEN_HOP) -> List[Tuple[int, int, bool]]:
    """[(visible tokens, token_offset, finalize)] of every flow call for an utterance whose LM produced n_tokens
    (cli/model.py:336-378; which calls are made does not depend on when the tokens arrive)."""
    pad = int(math.ceil(prompt_len / token_hop) * token_hop - prompt_len)
    calls, offset = [], 0
    while True:
        this_hop = token_hop + pad if offset == 0 else token_hop
        if n_tokens - offset >= this_hop + lookahead:
            calls.append((offset + this_hop + lookahead, offset, False))
            offset += this_hop
        else:
            break
    calls.append((n_tokens, offset, True))
    return calls


def stream_passes(flow_sd, tokens: torch.Tensor, prompt_token: torch.Tensor, prompt_feat: torch.Tensor,
                  embedding: torch.Tensor, ratio: int = 2):
    """[(token_offset * ratio, latents [E, 80] of ALL frames of the call, finalize)] of every flow call of the schedule
    (frames counted over the NEW tokens: flow.inference drops the prompt frames itself, flow.py:509).  The reference hands
    its vocoder the frames from token_offset * ratio on (:296); the DAC rule below also reads the call's own values of the
    ctx_right + MEL_CACHE frames before that point, which the previous pass could not finish."""
    out = []
    for vis, offset, fin in hop_schedule(tokens.shape[1], prompt_token.shape[1]):
        mel = OFLOW.flow_inference(flow_sd, tokens[:, :vis], prompt_token, prompt_feat, embedding,
                                   streaming=not fin, finalize=fin)        # closing pass: `stream` left at False
        out.append((offset * ratio, mel[0].t().contiguous(), fin))
    return out


def fade_window(n: int) -> torch.Tensor:
    """np.hamming(2n) (cli/model.py:262) with each pair (w[i], w[i + n]) scaled to sum to one."""
    w = torch.from_numpy(np.hamming(2 * n)).float()
    s = w[:n] + w[n:]
    return torch.cat([w[:n] / s, w[n:] / s])


def fade_in_out(fade_in, fade_out, window):
    """utils/common.py:142-150."""
    n = window.shape[0] // 2
    out = fade_in.clone()
    out[..., :n] = out[..., :n] * window[:n] + fade_out[..., -n:] * window[n:]
    return out


def render_passes(dac_sd, rates, passes, ctx_left: int, ctx_right: int):
    """The DAC rendering rule of the module docstring over [(token_offset * ratio, latents of all frames, finalize)] passes.
    Returns the emitted waveform chunks (1-D tensors, one per pass that emits something)."""
    hop = int(np.prod(rates))
    rendered = torch.zeros(0, 80)       # latent values each rendered frame was rendered from, by absolute frame
    held = None                         # samples of the last MEL_CACHE rendered frames, not yet emitted
    wavs = []
    for _, lat, fin in passes:
        end = lat.shape[0]
        hi = end if fin else end - ctx_right                 # frames whose right context is final in this pass
        lo = rendered.shape[0]                                # first frame not yet rendered
        if hi <= lo:
            continue
        re = held.shape[0] // hop if (fin and held is not None) else 0   # the closing pass renders the held frames again
        start = lo - re
        cl = min(ctx_left, start)
        win_lat = torch.cat([rendered[start - cl:start], lat[start:]], 0)
        wav = ODAC.decode(dac_sd, win_lat.t().unsqueeze(0).contiguous(), rates)[0, 0]
        seg = wav[cl * hop:(cl + hi - start) * hop]          # samples of frames [start, hi)
        if fin:
            out = fade_in_out(seg, held, fade_window(re * hop)) if re else seg
            held = None
        else:
            keep = min(MEL_CACHE, hi - lo)                    # (a pass renders at least MEL_CACHE frames in practice)
            out = seg[:seg.shape[0] - keep * hop]
            if held is not None:
                out = torch.cat([held, out])
            held = seg[seg.shape[0] - keep * hop:].clone()
        rendered = torch.cat([rendered, lat[lo:hi]], 0)
        if out.numel():
            wavs.append(out)
    return wavs


def tts_stream(flow_sd, dac_sd, rates, tokens, prompt_token, prompt_feat, embedding, ctx_left, ctx_right):
    """Waveform chunks of CosyVoice2Model.tts(stream=True) for given speech tokens (the LM side is oracle.llm)."""
    return render_passes(dac_sd, rates, stream_passes(flow_sd, tokens, prompt_token, prompt_feat, embedding), ctx_left, ctx_right)
