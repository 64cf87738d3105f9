"""End-to-end TTS hot path: text ids -> (LM) FSQ speech tokens -> (flow) DAC-VAE latents -> (DAC decoder) waveform.

This is the glue the reference leaves unwritten (speech/inference.py is empty): it follows
CosyVoice2Model.tts / token2wav (speech/cosyvoice/cli/model.py:285-386, non-streaming branch) with the HiFT
vocoder call (model.py:316) replaced by DACVAE.decode, as the README pipeline and the flow's training target
(speech_latent, flow.py:388-389) imply.
"""
import os
import time
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT, X2, X3, is_split
from .dac import DacDecoderEngine
from .flow import FlowEngine, Graphed
from .llm import LlmEngine

TOKEN_RATE = 25          # FSQ tokens per second (config.yaml:12)
SAMPLE_RATE = 24000


def fade_in_out(fade_in, fade_out, window):
    """speech/cosyvoice/utils/common.py:142-150 on device tensors: the head of `fade_in` is mixed with the tail of
    `fade_out` over the two halves of `window`."""
    n = window.shape[0] // 2
    out = fade_in.clone()
    out[..., :n] = out[..., :n] * window[:n] + fade_out[..., -n:] * window[n:]
    return out


class TtsEngine:
    def __init__(self, llm_sd, flow_sd, dac_sd, dtype=BF16, device="cuda", max_batch=1, max_ctx=2048,
                 dac_rates=(5, 4, 4, 3, 2), use_graphs=True, attn="bf16", wplanes=False):
        """wplanes (split build only): the weight-plane mode for checkpoints whose weights are NOT bf16-representable - what the
        reference's loaders hand over (fp32 llm.pt / flow.pt, cli/model.py:67-75; trained weight norms, dac-vae/inference.py:42-46).
        Every GEMM weight is carried as bf16 planes of its fp32 value (3 in the LM, 2 in the flow and the DAC: include/mmx_hip.h
        MMX_X3W / MMX_X2W) and the products keep every term above the last kept bit, so ids identical / waveform <= 1e-3 hold on
        such checkpoints too (tests/test_gpu_split.py::test_weight_planes_*)."""
        self.dtype, self.dev = dtype, torch.device(device)
        self.wplanes = wplanes if is_split(dtype) else False        # True / False / "auto" (per model: ops.resolve_wplanes)
        if self.dev.type == "cuda" and self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        device = self.dev
        # the split build: the flow and the DAC keep 16 significant bits of every activation (X2: waveform error ~1e-4),
        # the LM 24 (X3): its sampler turns a log-prob error into a different token id, and the AR loop feeds that back
        ldt = X3 if is_split(dtype) else dtype
        self.llm = LlmEngine(llm_sd, dtype=ldt, device=device, max_batch=max_batch, max_ctx=max_ctx, use_graphs=use_graphs,
                             wplanes=self.wplanes)
        # the decode step costs ~40 % more at 17..32 rows than at <= 16 (two MFMA row tiles): once at most 16
        # sequences are still running the batch continues in a 16-slot engine over the same weights and KV pages
        self.llm_small = (LlmEngine(None, dtype=ldt, device=device, max_batch=16, max_ctx=max_ctx, use_graphs=use_graphs,
                                    share_from=self.llm) if max_batch > 16 else None)
        self.flow = FlowEngine(flow_sd, dtype=dtype, device=device, use_graphs=use_graphs, attn=attn, wplanes=self.wplanes)
        self.dac = DacDecoderEngine(dac_sd, list(dac_rates), dtype=dtype, device=device, wplanes=self.wplanes)
        self.hop = self.dac.hop
        # tts_batch's cost model of its own stages (ms): a decode step beside the flow; a flow group = group_ms + frame_ms per frame.
        # Initial values: what _refit_sched measures for this build on one MI355X on the config-4 share (it re-measures them in
        # every call: self.sched_fit; sched_adapt lets the model follow the fit)
        self.sched = ({"step_ms": 1.05, "group_ms": 47.0, "frame_ms": 0.026} if is_split(dtype) else
                      {"step_ms": 0.84, "group_ms": 35.0, "frame_ms": 0.009})
        self.sched_fit, self._sched_prev = None, None

    # auxiliary streams per flow group for its per-utterance stages (conformer encoder, DAC decode); 1 = off.  Off by default:
    # measured on the config-4 rank share, 2 / 3 / 4 streams cost 15 % / 13 % / 32 % of the step (more queues beside the decode
    # loop slow its launches more than the per-utterance stages gain); useful only without a decode loop alongside
    group_fan = 1
    batch_encoder = True      # the conformer encoder of a flow group runs as one zero-padded batch (FlowEngine.encode_batch)
    batch_dac = True          # ... and so does its DAC decode (DacDecoderEngine.decode_time_major with per-member lengths)

    flow_priority = 0      # HIP stream priority of the flow workers' streams (the decode loop runs at -1 = high)

    def _flow_stream(self):
        """A flow worker's stream.  Priorities outside torch's range (HIP has a low level, +1) go through
        hipStreamCreateWithPriority and torch.cuda.ExternalStream."""
        if self.flow_priority in (0, -1):
            return torch.cuda.Stream(device=self.dev, priority=self.flow_priority)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        h = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(0), ctypes.c_int(self.flow_priority))
        if rc != 0:
            raise RuntimeError(f"hipStreamCreateWithPriority({self.flow_priority}) -> {rc}")
        self.__dict__.setdefault("_ext_streams", []).append(h)
        return torch.cuda.ExternalStream(h.value, device=self.dev)

    def _aux_streams(self, cur, n):
        """n auxiliary streams belonging to the stream `cur` (one set per flow worker stream; created once)."""
        pool = self.__dict__.setdefault("_aux", {})
        key = cur.cuda_stream
        if key not in pool or len(pool[key]) < n:
            pool[key] = [torch.cuda.Stream(device=self.dev, priority=0) for _ in range(n)]
        return pool[key][:n]

    def close(self):
        """Destroys every recorded hipGraph of the engines on the calling thread (deterministic teardown: nothing is left
        for Python's cyclic collector to finalise later on whatever thread it happens to run)."""
        self.llm.close()
        if self.llm_small is not None:
            self.llm_small.close()
        for fl in getattr(self, "_flows", [self.flow]):
            fl.close()
        self.flow.close()

    @torch.no_grad()
    def generate_tokens(self, texts: List[torch.Tensor], prompt_texts=None, prompt_speech=None, seed=0,
                        min_ratio=2, max_ratio=20, exact_steps=None) -> List[torch.Tensor]:
        """Batched AR decode (Qwen2LM.inference semantics per sequence). exact_steps (int or list) forces exactly
        that many sampling steps with EOS ignored (BASELINE config 3: 250 steps for a 10 s utterance)."""
        B = len(texts)
        assert B == self.llm.B
        if exact_steps is not None and not isinstance(exact_steps, (list, tuple)):
            exact_steps = [exact_steps] * B
        z = lambda: torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        xs, mins, maxs = [], [], []
        for b in range(B):
            pt = prompt_texts[b] if prompt_texts else z()
            ps = prompt_speech[b] if prompt_speech else z()
            xs.append(self.llm.build_lm_input(texts[b], pt, ps))
            n = texts[b].numel()
            mins.append(exact_steps[b] if exact_steps is not None else int(n * min_ratio))
            maxs.append(exact_steps[b] if exact_steps is not None else int(n * max_ratio))
        self.llm.start(xs, mins, maxs, seed=seed)
        self.llm.run(max(maxs))
        n = self.llm.state[2].tolist()
        return [self.llm.out_tokens[b, :n[b]].to(torch.int64) for b in range(B)]

    @torch.no_grad()
    def token2wav(self, token: torch.Tensor, prompt_token: torch.Tensor, prompt_feat: torch.Tensor,
                  embedding: torch.Tensor, streaming=False, finalize=True) -> torch.Tensor:
        """tokens [1,L] -> waveform [1, 1, 2*L*hop] (cli/model.py:285-319 with the vocoder swapped for DAC)."""
        lat = self.flow.inference_time_major(token, prompt_token, prompt_feat, embedding, streaming, finalize)
        T2 = lat.shape[0]
        zt = torch.empty(1, T2, 80, dtype=TORCH_DT[self.dtype], device=self.dev)
        ops.copy2d(lat, F32, 0, 80, 1, zt, self.dtype, 0, 80, 1, rows=T2, cols=80)
        return self.dac.decode_time_major(zt, 1, T2)

    @torch.no_grad()
    def tts(self, text, flow_embedding, prompt_text=None, llm_prompt_speech_token=None, flow_prompt_speech_token=None,
            prompt_speech_feat=None, seed=0, exact_steps=None) -> torch.Tensor:
        """One utterance, non-streaming (cli/model.py:321-386 `stream=False` branch)."""
        z = torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        toks = self.generate_tokens([text], [prompt_text if prompt_text is not None else z],
                                    [llm_prompt_speech_token if llm_prompt_speech_token is not None else z],
                                    seed=seed, exact_steps=exact_steps)[0]
        pf = prompt_speech_feat if prompt_speech_feat is not None else torch.zeros(1, 0, 80, device=self.dev)
        pt = flow_prompt_speech_token if flow_prompt_speech_token is not None else z
        return self.token2wav(toks.reshape(1, -1), pt, pf, flow_embedding)

    MEL_CACHE = 8            # cli/model.py:258: frames of overlap between two passes

    @staticmethod
    def fade_window(n: int, device) -> torch.Tensor:
        """np.hamming(2n) (cli/model.py:262) with each pair (w[i], w[i + n]) scaled to sum to one: a cross-fade of two equal
        signals is then the identity (the raw halves sum to 1.08)."""
        import numpy as np
        w = torch.from_numpy(np.hamming(2 * n)).float()
        s_ = w[:n] + w[n:]
        return torch.cat([w[:n] / s_, w[n:] / s_]).to(device)

    @torch.no_grad()
    def tts_stream(self, text, flow_embedding, seed=0, exact_steps=None, token_hop=25, latents_out=None, forced=None,
                   cache=True, prompt_text=None, llm_prompt_speech_token=None, flow_prompt_speech_token=None,
                   prompt_speech_feat=None):
        """Streaming synthesis of one (long) utterance: BASELINE config 5 / cli/model.py:336-378 (`stream=True`), zero-shot
        prompts included (prompt_text / llm_prompt_speech_token condition the LM, llm.py:691-703; flow_prompt_speech_token /
        prompt_speech_feat the flow, flow.py:472-498).
        The AR decode runs ahead on its own stream (captured decode step).  Hop schedule as the reference: the first hop takes
        token_hop + prompt_token_pad tokens (pad = ceil(Lp / 25) * 25 - Lp, model.py:338-341), later hops token_hop; a hop runs
        once `hop + look-ahead` tokens exist and solves the chunk-causal flow over the tokens so far; the closing pass solves
        everything WITHOUT chunk masks (model.py:371-378 leaves `stream` at False).
        Rendering (the DAC decoder replaces HiFT and its mel / source caches, model.py:298-311; rule restated in
        oracle/stream.py): the DAC is NOT causal — a sample depends on `dac.ctx_left` latent frames before and `dac.ctx_right`
        after its own frame — so a pass renders the frames whose right context is final, from a window with ctx_left frames of
        ALREADY RENDERED left context; streaming passes agree on finished frames, so their chunks equal the offline decode of
        the same latents sample for sample.  Like the reference (model.py:306-311) a streaming pass holds the samples of its
        last MEL_CACHE frames back: the next streaming pass emits them unchanged, the CLOSING pass — the one seam where two
        passes disagree about the frames around it — renders those frames again from its own latents and cross-fades the two
        renderings (utils/common.py:142-150 fade_in_out, window of model.py:262 normalised to unit sum).
        Yields waveform chunks [1, n] (device); `latents_out` (a list) receives the latent frames [n, 80] each chunk was
        rendered from; `forced` [1, steps] teacher-forces the accepted ids (LlmEngine.start).  cache=True: hops solve only
        their new frames (FlowEngine.StreamState); cache=False recomputes all frames at every hop, as the reference does."""
        from .llm import ST_FIN, ST_NOUT
        assert self.llm.B == 1
        z = torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        zf = torch.zeros(1, 0, 80, device=self.dev)
        pt = prompt_text if prompt_text is not None else z
        lps = llm_prompt_speech_token if llm_prompt_speech_token is not None else z
        fpt = flow_prompt_speech_token if flow_prompt_speech_token is not None else z
        pf = prompt_speech_feat if prompt_speech_feat is not None else zf
        x = self.llm.build_lm_input(text, pt, lps)
        n_text = int(text.numel())
        mn = exact_steps if exact_steps is not None else n_text * 2
        mx = exact_steps if exact_steps is not None else n_text * 20
        if not hasattr(self, "_lm_stream"):
            self._lm_stream = torch.cuda.Stream(device=self.dev, priority=-1)
        lm, caller = self._lm_stream, torch.cuda.current_stream()
        lm.wait_stream(caller)
        L = self.flow.L
        CL, CR, MC = self.dac.ctx_left, self.dac.ctx_right, self.MEL_CACHE
        Lp = int(fpt.numel())
        pad = -(-Lp // token_hop) * token_hop - Lp              # model.py:338
        with torch.cuda.stream(lm):
            self.llm.start([x], [mn], [mx], seed=seed, forced=forced)
        done, offset = 1, 0                                  # decode steps issued, tokens consumed by hops
        emitted = 0                                          # latent frames rendered
        sstate = self.flow.stream_open(2 * (mx + Lp)) if cache else None
        tail = None                                          # the last ctx_left + MEL_CACHE rendered latent frames
        held = None                                          # samples of the last MEL_CACHE rendered frames, not yet emitted
        held_lat = None

        def render(n_tok, finalize):
            nonlocal emitted, tail, held, held_lat
            ev = torch.cuda.Event()
            ev.record(lm)
            caller.wait_event(ev)                          # tokens [0, n_tok) are written
            tok = self.llm.out_tokens[0:1, :n_tok].to(torch.int64)
            lat = self.flow.inference_time_major(tok, fpt, pf, flow_embedding, streaming=not finalize, finalize=finalize,
                                                 stream_state=sstate)
            T2 = lat.shape[0]
            hi = T2 if finalize else T2 - CR               # frames whose right context is final
            if hi <= emitted:
                return None
            # the closing pass renders the held frames again (as many as the last streaming pass held back: MEL_CACHE, or all
            # it rendered when that was fewer)
            re = held.shape[1] // self.hop if (finalize and held is not None) else 0
            start = emitted - re
            ctx = None if tail is None else tail[:tail.shape[0] - re][-CL:]
            seg = lat[start:] if ctx is None or ctx.shape[0] == 0 else torch.cat([ctx, lat[start:]], dim=0)
            nctx = 0 if ctx is None else ctx.shape[0]
            n = seg.shape[0]
            zt = torch.empty(1, n, 80, dtype=TORCH_DT[self.dtype], device=self.dev)
            ops.copy2d(seg.contiguous(), F32, 0, 80, 1, zt, self.dtype, 0, 80, 1, rows=n, cols=80)
            wav = self.dac.decode_time_major(zt, 1, n)[:, 0, nctx * self.hop:(nctx + hi - start) * self.hop]
            lat_used = lat[emitted:hi]
            if finalize:
                if re:
                    wav = fade_in_out(wav, held, self.fade_window(re * self.hop, self.dev))
                    lat_used = torch.cat([held_lat, lat_used], 0)
                out, held, held_lat = wav, None, None
            else:
                keep = min(MC, hi - emitted)
                out = wav[:, :wav.shape[1] - keep * self.hop]
                if held is not None:
                    out = torch.cat([held, out], dim=1)
                    lat_used = torch.cat([held_lat, lat_used], 0)
                held = wav[:, wav.shape[1] - keep * self.hop:].clone()
                held_lat, lat_used = lat_used[lat_used.shape[0] - keep:].clone(), lat_used[:lat_used.shape[0] - keep]
            if latents_out is not None and lat_used.shape[0]:
                latents_out.append(lat_used.clone())
            new = lat[emitted:hi]
            tail = (new if tail is None else torch.cat([tail, new], 0))[-(CL + MC):].clone()
            emitted = hi
            return out if out.shape[1] else None

        try:
            while True:
                with torch.cuda.stream(lm):
                    st = self.llm.state[:, 0].tolist()         # D2H copy on the LM stream: waits for the steps issued so far
                n_out, finished = st[ST_NOUT], bool(st[ST_FIN]) or done >= mx
                this_hop = token_hop + pad if offset == 0 else token_hop          # model.py:341
                while n_out - offset >= this_hop + L:
                    w = render(offset + this_hop + L, finalize=False)
                    offset += this_hop
                    this_hop = token_hop
                    if w is not None:
                        yield w
                if finished:
                    w = render(n_out, finalize=True)
                    if w is not None:
                        yield w
                    break
                # keep the decode a few ACCEPTED tokens ahead of the renderer.  The look-ahead is counted in tokens, not in
                # steps (ids above the EOS id advance the step counter without producing a token, llm.py:755-756), and at
                # least one step is issued per round, so the loop always makes progress.
                k = min(mx - done, max(1, offset + this_hop + L + 8 - n_out))
                with torch.cuda.stream(lm):
                    for _ in range(k):
                        self.llm.step()
                done += k
        finally:
            if sstate is not None:
                self.flow.stream_close(sstate)           # the state (and its hop graphs) goes back to the pool
        caller.wait_stream(lm)

    # ------------------------------------------------------------------ batch of independent utterances
    def _groups(self, order, frames, group_size, max_pad_ratio, frame_quantum, first=0):
        """Consecutive runs of `order` (sorted by length) whose lengths are within the padding budget.
        group_size may be a list: the size limit of the k-th group issued (k counted from `first`); the last entry
        repeats.  A ramp such as [2, 2, 4, 8] lets the flow stage start as soon as the two shortest utterances are
        decoded instead of waiting for eight."""
        sizes = group_size if isinstance(group_size, (list, tuple)) else [group_size]
        out, i = [], 0
        while i < len(order):
            gs = sizes[min(first + len(out), len(sizes) - 1)]
            j, t0 = i + 1, frames[order[i]]
            while j < len(order) and j - i < gs and frames[order[j]] <= max(t0 * max_pad_ratio, t0 + frame_quantum):
                j += 1
            out.append(order[i:j])
            i = j
        return out

    def _flow_dac_group(self, grp, toks, embs, wavs, frame_quantum, flow=None, prompts=None):
        """Flow + DAC for a group of finished utterances: per-utterance conformer encoder, one batched ODE solve, per-
        utterance DAC decode.  prompts: per utterance (flow_prompt_speech_token [1, Lp], prompt_speech_feat [1, Tp, 80]) or
        None (flow.py:472-498: the prompt's tokens go through the encoder in front of the utterance's, its latents are the
        `cond` rows of its frames, and its frames are dropped from the result).
        MMX_TIMING=3 prints the three stage times (with stream syncs between them)."""
        flow = flow or self.flow
        z = torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        zf = torch.zeros(1, 0, 80, device=self.dev)
        pr = lambda b: (z, zf) if (prompts is None or prompts[b] is None) else prompts[b]
        trace = os.environ.get("MMX_TIMING") == "3"
        marks = []

        def mark():
            if trace:
                torch.cuda.current_stream().synchronize()
                marks.append(time.perf_counter())

        # The conformer encoder and the DAC decode are per utterance: chains of ~100 / ~30 launches, most of them too small to
        # fill the chip.  The utterances of a group are independent there, so each goes to one of `fan` auxiliary streams
        # (forked from / joined into the group's stream with events); the batched ODE solve between them stays on the group's
        # stream.  fan = 1: everything on the group's stream.
        cur = torch.cuda.current_stream()
        fan = min(self.group_fan, len(grp))
        aux = self._aux_streams(cur, fan) if fan > 1 else []

        def fanned(jobs):
            """jobs: callables, one per utterance; run on the auxiliary streams round robin, joined before returning."""
            if not aux:
                return [j() for j in jobs]
            ev0 = torch.cuda.Event()
            ev0.record(cur)
            out = []
            for i, j in enumerate(jobs):
                st = aux[i % fan]
                st.wait_event(ev0)
                with torch.cuda.stream(st):
                    out.append(j())
            for st in aux:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            for o in out:                                    # results live on after this call, on other streams
                for t in (o if isinstance(o, tuple) else (o,)):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(cur)
            return out

        mark()
        if self.batch_encoder and len(grp) > 1:
            # one encoder pass over the zero-padded group (FlowEngine.encode_batch)
            conds = flow.conditions_batch([toks[b].reshape(1, -1) for b in grp], [pr(b)[0] for b in grp], [pr(b)[1] for b in grp],
                                          [embs[b] for b in grp])
        else:
            conds = fanned([(lambda b=b: flow.conditions(toks[b].reshape(1, -1), pr(b)[0], pr(b)[1], embs[b])) for b in grp])
        mark()
        xs = flow.cfm_batch([c[0] for c in conds], [c[1] for c in conds], [c[2] for c in conds], pad_to=frame_quantum)
        mark()

        def dac_job(b, lat, c):
            lat = lat[c[3]:]                                 # the prompt's frames are not rendered (flow.py:509)
            T2 = lat.shape[0]
            zt = torch.empty(1, T2, 80, dtype=TORCH_DT[self.dtype], device=self.dev)
            ops.copy2d(lat, F32, 0, 80, 1, zt, self.dtype, 0, 80, 1, rows=T2, cols=80)
            return self.dac.decode_time_major(zt, 1, T2)

        if self.batch_dac and len(grp) > 1 and not aux:
            # ONE decode of the zero-padded group (DacDecoderEngine.decode_time_major with per-member lengths: row masks in the
            # GEMM epilogues, lengths in the fused ResidualUnits) instead of ~32 launches per utterance; member i of the result
            # equals its own decode bit for bit
            lats = [lat[c[3]:] for lat, c in zip(xs, conds)]
            Ts = [int(l.shape[0]) for l in lats]
            Tm = max(Ts)
            zt = torch.zeros(len(grp), Tm, 80, dtype=TORCH_DT[self.dtype], device=self.dev)
            for i, l in enumerate(lats):
                ops.copy2d(l, F32, 0, 80, 1, zt[i], self.dtype, 0, 80, 1, rows=Ts[i], cols=80)
            wav = self.dac.decode_time_major(zt, len(grp), Tm, lens=Ts)
            for i, b in enumerate(grp):
                wavs[b] = wav[i:i + 1, :, :Ts[i] * self.hop]
        else:
            for b, w in zip(grp, fanned([(lambda b=b, lat=lat, c=c: dac_job(b, lat, c)) for b, lat, c in zip(grp, xs, conds)])):
                wavs[b] = w
        if aux:
            # blocks the auxiliary streams allocated (and this function's temporaries freed there) go back to THEIR pools:
            # nothing on them may be reused before the group's stream has finished reading it
            ev = torch.cuda.Event()
            ev.record(cur)
            for st in aux:
                st.wait_event(ev)
        mark()
        if trace:
            e, c, d = ((marks[i + 1] - marks[i]) * 1e3 for i in range(3))
            print(f"[flow_dac_group] n={len(grp)} frames={[2 * toks[b].numel() for b in grp]}: encoder {e:.1f} ms, "
                  f"cfm {c:.1f} ms, dac {d:.1f} ms", flush=True)

    @torch.no_grad()
    def tts_batch(self, texts, flow_embeddings, seed=0, exact_steps=None, group_size=8, max_pad_ratio=2.0,
                  frame_quantum=32, overlap=True, poll_every=8, flow_workers=2, hold_steps=60, tail_active=0, polite=True,
                  prompt_texts=None, llm_prompt_speech_tokens=None, flow_prompt_speech_tokens=None,
                  prompt_speech_feats=None) -> List[torch.Tensor]:
        """Throughput path for a batch of independent utterances (BASELINE config 4, one rank's share): one batched
        AR decode for all of them; as sequences finish (shortest first) their flow + DAC work — per-utterance
        conformer encoder, ODE solves batched over groups of similar length (zero padded + masked), DAC decode — is
        issued by a second host thread on a second HIP stream, so the latency-bound decode loop and the MFMA-bound
        flow overlap on the chip (the reference overlaps the same two stages with its llm_job thread,
        cli/model.py:332-335).  overlap=False runs the stages back to back.
        Zero-shot prompts (cli/cosyvoice.py:92-104 -> cli/model.py:321-326), each a per-utterance list or None: prompt_texts
        and llm_prompt_speech_tokens condition the LM (llm.py:691-703), flow_prompt_speech_tokens / prompt_speech_feats the
        flow (flow.py:472-498).
        group_size / hold_steps: a finished utterance waits at most hold_steps decode steps for up to group_size companions
        of similar length.  Large groups pay twice: a launch of the fused flow kernels costs about the same from 500 to
        8 000 rows (it is bound by every workgroup streaming the block's weights), so fewer, fuller groups are less GPU
        work, and less flow work beside the decode loop makes the decode loop itself faster (measured, 32 utterances:
        ramp 2,2,4,8 / hold 40: 718 ms per step, decode loop 617 ms; groups of 8 / hold 60: 703 ms, decode loop 577 ms);
        then its (partial) group is issued, so the flow work of the long utterances is not left for after the last
        token (the rule counts decode steps, not wall time: the schedule, and with it the set of captured plans, is
        the same from run to run).
        polite: flow groups issued while the decode loop runs use FlowEngine.polite tiling (64-row tiles: fewer workgroups,
        more of the chip left to the decode loop's launches); the groups of the final harvest use the fastest tiling.
        tail_active > 0: once at most that many sequences are still decoding, a finished utterance no longer waits for
        companions when a flow worker is (predicted) idle - the decode loop is the critical path, and whatever the last
        utterances still have to do after their last token is what the step ends on."""
        import queue as queue_mod
        import threading
        from .llm import ST_FIN, ST_NOUT, ST_POS
        B = len(texts)
        NS = self.llm.B                                   # decode slots; more utterances than slots queue up and are admitted
        assert B >= NS and (overlap or B == NS)           # into slots as they free (continuous batching, LlmEngine.admit)
        if exact_steps is not None and not isinstance(exact_steps, (list, tuple)):
            exact_steps = [exact_steps] * B
        z = torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        pick = lambda lst, b: z if (lst is None or lst[b] is None) else lst[b]
        xs = [self.llm.build_lm_input(t, pick(prompt_texts, b), pick(llm_prompt_speech_tokens, b)) for b, t in enumerate(texts)]
        prompts = None
        if flow_prompt_speech_tokens is not None:
            zf = torch.zeros(1, 0, 80, device=self.dev)
            prompts = [None if flow_prompt_speech_tokens[b] is None else
                       (flow_prompt_speech_tokens[b], zf if prompt_speech_feats is None or prompt_speech_feats[b] is None else prompt_speech_feats[b])
                       for b in range(B)]
        plen = [0 if (prompts is None or prompts[b] is None) else int(prompts[b][0].numel()) for b in range(B)]   # prompt tokens in the flow
        mins = [exact_steps[b] if exact_steps is not None else int(texts[b].numel() * 2) for b in range(B)]
        maxs = [exact_steps[b] if exact_steps is not None else int(texts[b].numel() * 20) for b in range(B)]
        wavs: List[Optional[torch.Tensor]] = [None] * B
        toks: List[Optional[torch.Tensor]] = [None] * B
        if not overlap:
            self.llm.start(xs, mins, maxs, seed=seed)
            self.llm.run(max(maxs), poll_every)
            n = self.llm.state[ST_NOUT].tolist()
            for b in range(B):
                toks[b] = self.llm.out_tokens[b, :n[b]].to(torch.int64)
            order = sorted(range(B), key=lambda b: (n[b], b))
            for grp in self._groups(order, [2 * (v + plen[b]) for b, v in enumerate(n)], group_size, max_pad_ratio, frame_quantum):
                self._flow_dac_group(grp, toks, flow_embeddings, wavs, frame_quantum, prompts=prompts)
            self.last_tokens = toks
            return wavs

        if not hasattr(self, "_sides") or len(self._sides) != flow_workers:
            # the decode loop is a chain of short latency-bound kernels: give it the high-priority queue so its
            # launches are not parked behind the flow's large grids.  The flow stage itself is a chain of short
            # kernels too, so `flow_workers` host threads, each with its own stream and its own plan buffers (the
            # weights are shared), solve different groups concurrently.
            # (CU-masked flow streams, hipExtStreamCreateWithCUMask, were tried to keep CUs free for the decode loop:
            # the mask is not honoured on this pool — an 8192^3 GEMM takes the same time with 1/4 and 4/4 of the CUs)
            self._sides = [self._flow_stream() for _ in range(flow_workers)]
            self._flows = [self.flow] + [self.flow.clone_shared() for _ in range(flow_workers - 1)]
            self._hi = torch.cuda.Stream(device=self.dev, priority=-1)
        qs, err = [queue_mod.Queue() for _ in range(flow_workers)], []
        caller = torch.cuda.current_stream()
        self._hi.wait_stream(caller)

        def worker(wi):
            side, flow = self._sides[wi], self._flows[wi]
            try:
                torch.cuda.set_device(self.dev)
                with torch.cuda.stream(side):
                    while True:
                        item = qs[wi].get()
                        if item is None:
                            return
                        grp, ev, flow.polite = item
                        side.wait_event(ev)                      # the group's token ids were written on the LM stream
                        t_in = _time.perf_counter()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(side)
                        self._flow_dac_group(grp, toks, flow_embeddings, wavs, frame_quantum, flow, prompts)
                        e1.record(side)                          # read after the call has drained: the scheduler's cost model
                        group_events.append((sum(2 * toks[b].numel() for b in grp), flow.polite, e0, e1))
                        if _trace:
                            side.synchronize()
                            print(f"[tts_batch]   worker {wi}: group of {len(grp)} ({[2 * toks[b].numel() for b in grp]} frames) "
                                  f"{(t_in - self._t0) * 1e3:.0f} -> {(_time.perf_counter() - self._t0) * 1e3:.0f} ms", flush=True)
            except BaseException as e:                           # surfaced by the caller
                err.append(e)

        _os, _time = os, time
        _trace = os.environ.get("MMX_TIMING") == "2"
        self._t0 = _time.perf_counter()
        ths = [threading.Thread(target=worker, args=(wi,), daemon=True) for wi in range(flow_workers)]
        for th in ths:
            th.start()
        main = self._hi
        pending: List[int] = []
        seen = set()
        issued = [0]
        arrived, steps_done = {}, [1]
        free_at = [0.0] * flow_workers
        # the scheduler's cost model (self.sched): a decode step STEP_MS; a flow group GROUP_MS + FRAME_MS per frame - measured by
        # every call for itself (_refit_sched) and, with sched_adapt, followed when two steady calls in a row say it is off by more
        # than SCHED_HYSTERESIS (a steady workload keeps one assignment - and one set of captured plans - from call to call)
        STEP_MS, GROUP_MS, FRAME_MS = self.sched["step_ms"], self.sched["group_ms"], self.sched["frame_ms"]
        group_events: list = []
        cold0 = Graphed.cold_calls

        cur = [self.llm, list(range(NS))]                           # active engine, slot -> utterance index
        waiting = list(range(NS, B))                                # utterances waiting for a slot

        def harvest(final):
            eng, slots = cur
            fin = eng.state[ST_FIN].tolist()
            n = eng.state[ST_NOUT].tolist()
            new = sorted([s_ for s_ in range(len(slots)) if (fin[s_] or final) and slots[s_] not in seen],
                         key=lambda s_: (n[s_], slots[s_]))
            for s_ in new:
                b = slots[s_]
                seen.add(b)
                toks[b] = eng.out_tokens[s_, :n[s_]].to(torch.int64)
                pending.append(b)
                arrived[b] = steps_done[0]
            while waiting and not final:                            # a freed slot takes the next queued utterance
                free = [s_ for s_ in range(len(slots)) if fin[s_] and slots[s_] in seen and slots[s_] >= 0]
                if not free:
                    break
                s_, b = free[0], waiting.pop(0)
                eng.admit(s_, xs[b], mins[b], maxs[b], seq_id=b)    # reserves KV pages for the whole max_len
                slots[s_] = b
                fin[s_] = 0
            if not final and eng is self.llm:
                # every running sequence must own the pages the steps up to the next poll will write (a no-op when admit /
                # start reserved the whole max_len; raises when the allocator is exhausted instead of decoding into the
                # shared scratch page)
                eng.ensure_capacity(poll_every + 1, pos=eng.state[ST_POS].tolist(),
                                    active=[s_ for s_ in range(len(slots)) if not fin[s_]])
            if (not final and not waiting and self.llm_small is not None and eng is self.llm and B - len(seen) <= self.llm_small.B
                    and B - len(seen) > 0):
                act = [s_ for s_ in range(len(slots)) if slots[s_] not in seen]
                self.llm_small.compact_from(self.llm, act)
                cur[0], cur[1] = self.llm_small, [slots[s_] for s_ in act]
            frames = {b: 2 * (toks[b].numel() + plen[b]) for b in pending}
            groups = self._groups(pending, frames, group_size, max_pad_ratio, frame_quantum, first=issued[0])
            sizes = group_size if isinstance(group_size, (list, tuple)) else [group_size]
            now = steps_done[0] * STEP_MS
            if not final and groups:
                want = sizes[min(issued[0] + len(groups) - 1, len(sizes) - 1)]
                waited = steps_done[0] - min(arrived[b] for b in groups[-1])
                rush = 0 < B - len(seen) <= tail_active and any(free_at[w] <= now for w in range(flow_workers))
                if len(groups[-1]) < want and not (hold_steps > 0 and waited >= hold_steps) and not rush:
                    groups = groups[:-1]                         # keep a partial group open for later arrivals
            assign = None
            if final and hold_steps > 0 and len(groups) == 1 and len(groups[0]) >= 2:
                # last arrivals: longest first, each to the worker predicted to finish it first (the other worker may
                # still be busy with an earlier group, then splitting only delays the end)
                fa = [max(free_at[w], now) for w in range(flow_workers)]
                parts = [[] for _ in range(flow_workers)]
                for b in sorted(groups[0], key=lambda b: -frames[b]):
                    w = min(range(flow_workers), key=lambda w: (fa[w] + (0.0 if parts[w] else GROUP_MS) + FRAME_MS * frames[b], w))
                    fa[w] += (0.0 if parts[w] else GROUP_MS) + FRAME_MS * frames[b]
                    parts[w].append(b)
                assign = [w for w in range(flow_workers) if parts[w]]
                groups = [sorted(parts[w], key=lambda b: (frames[b], b)) for w in assign]
            for grp in groups:
                ev = torch.cuda.Event()
                ev.record(main)
                # the worker predicted to be free first.  Prediction, not wall time: decode steps are the clock and a
                # group costs GROUP_MS + FRAME_MS per frame (fitted to MMX_TIMING=2 traces), so the assignment - and
                # with it every worker's set of captured plans - repeats from run to run
                wi = assign.pop(0) if assign else min(range(flow_workers), key=lambda w: (max(free_at[w], now), w))
                free_at[wi] = max(free_at[wi], now) + GROUP_MS + FRAME_MS * sum(frames[b] for b in grp)
                # groups issued while the decode loop is running use the flow kernels' polite tiling (FlowEngine.polite);
                # the last arrivals, issued when it has ended, the fastest one
                qs[wi].put((grp, ev, polite and not final and B == NS))    # (with a queue the flow stage is the bottleneck)
                issued[0] += 1
                for b in grp:
                    pending.remove(b)

        with torch.cuda.stream(main):
            self.llm.start(xs[:NS], mins[:NS], maxs[:NS], seed=seed)   # (captures serialise themselves: mmx/flow.py, Graphed)
            # without a queue max(maxs) steps end every sequence; with one, admissions happen only at polls, so the loop
            # runs until every utterance has been seen (the bound only stops a runaway: each admitted utterance can wait
            # up to poll_every steps for its slot on top of its own max_len)
            done, max_steps = 1, (max(maxs) if B == NS else sum(maxs) + (B + 1) * poll_every)
            issue_s = 0.0                                        # host time spent enqueueing decode steps (graph replays)
            while done < max_steps:
                k = min(poll_every, max_steps - done)
                t_i = _time.perf_counter()
                for _ in range(k):
                    cur[0].step()
                issue_s += _time.perf_counter() - t_i
                done += k
                steps_done[0] = done
                harvest(False)
                if len(seen) == B:
                    break
            if waiting:
                raise RuntimeError(f"tts_batch: {len(waiting)} queued utterances were never admitted")
            harvest(True)
        timing = os.environ.get("MMX_TIMING")
        # The call returns finished audio, so it drains its streams on the host as well: the decode stream here, the
        # flow streams after the workers have issued everything.  Leaving the drain to stream waits (so that the next
        # call's decode loop is already queued behind them) measured 6 % slower per step: a blocked high-priority queue
        # ahead of the still running flow tail costs the tail more than the host round trip saves.
        main.synchronize()
        t_lm = time.perf_counter()
        for qq in qs:
            qq.put(None)
        for th in ths:
            th.join()
        if err:
            raise err[0]
        for sd in self._sides:
            sd.synchronize()
        for fl in self._flows:
            fl.polite = False
        if timing:
            print(f"[tts_batch] LM loop done at {(t_lm - self._t0) * 1e3:.0f} ms, flow/DAC tail until {(time.perf_counter() - self._t0) * 1e3:.0f} ms, "
                  f"decode steps {done}", flush=True)
        for sd in self._sides:
            caller.wait_stream(sd)
        caller.wait_stream(main)
        self.last_tokens = toks                          # accepted ids per utterance (device int64 tensors)
        # host-side accounting of the call (bench.py puts it into its JSON line: a multi-GPU node runs N x (decode thread + flow
        # workers) on shared cores, and a slow host shows up here first)
        self.last_host = dict(decode_steps=done, lm_issue_ms=round(issue_s * 1e3, 2), lm_done_ms=round((t_lm - self._t0) * 1e3, 1),
                              call_ms=round((time.perf_counter() - self._t0) * 1e3, 1))
        self._refit_sched(done, (t_lm - self._t0) * 1e3, group_events, steady=(Graphed.cold_calls == cold0 and B == NS))
        return wavs

    SCHED_HYSTERESIS = 0.25
    sched_adapt = False       # True: self.sched follows the measured fit (off by default: a change of the model moves groups between
                              # workers, and a worker that meets a new group shape spends two calls capturing a plan for it)

    def _refit_sched(self, steps, lm_ms, group_events, steady=True):
        """The scheduler's cost model from this call's own clock: decode step = loop time / steps; group cost = least squares of
        (frames, HIP-event duration) over the groups that ran beside the decode loop.  self.sched_fit always holds the latest fit;
        with sched_adapt, self.sched (what the next call uses) follows it past SCHED_HYSTERESIS when two steady calls in a row gave the
        same fit, one parameter at a time."""
        if not steady:                                   # a call that captured plans (or queued utterances) times something else
            self._sched_prev = None
            return
        fit = {"step_ms": lm_ms / max(1, steps)}
        pts = [(f, e0.elapsed_time(e1)) for f, pol, e0, e1 in group_events if pol]
        if len(pts) >= 3 and len({f for f, _ in pts}) >= 2:
            n = float(len(pts))
            sx, sy = sum(f for f, _ in pts), sum(t for _, t in pts)
            sxx, sxy = sum(f * f for f, _ in pts), sum(f * t for f, t in pts)
            den = n * sxx - sx * sx
            slope = (n * sxy - sx * sy) / den if den > 0 else 0.0
            if slope > 0 and (sy - slope * sx) / n > 0:
                fit["frame_ms"], fit["group_ms"] = slope, (sy - slope * sx) / n
        prev, self._sched_prev = self._sched_prev, fit
        self.sched_fit = {k: round(v, 5) for k, v in fit.items()}
        if not self.sched_adapt:
            return
        for k, v in fit.items():
            if prev and k in prev and abs(v - prev[k]) <= 0.1 * prev[k] and abs(v - self.sched[k]) > self.SCHED_HYSTERESIS * self.sched[k]:
                self.sched[k] = round(v, 5)
