"""End-to-end TTS hot path: text ids -> (LM) FSQ speech tokens -> (flow) DAC-VAE latents -> (DAC decoder) waveform.

This is the glue the reference leaves unwritten (speech/inference.py is empty): it follows
CosyVoice2Model.tts / token2wav (speech/cosyvoice/cli/model.py:285-386, non-streaming branch) with the HiFT
vocoder call (model.py:316) replaced by DACVAE.decode, as the README pipeline and the flow's training target
(speech_latent, flow.py:388-389) imply.
"""
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT
from .dac import DacDecoderEngine
from .flow import FlowEngine
from .llm import LlmEngine

TOKEN_RATE = 25          # FSQ tokens per second (config.yaml:12)
SAMPLE_RATE = 24000


class TtsEngine:
    def __init__(self, llm_sd, flow_sd, dac_sd, dtype=BF16, device="cuda", max_batch=1, max_ctx=2048,
                 dac_rates=(5, 4, 4, 3, 2), use_graphs=True):
        self.dtype, self.dev = dtype, torch.device(device)
        self.llm = LlmEngine(llm_sd, dtype=dtype, device=device, max_batch=max_batch, max_ctx=max_ctx, use_graphs=use_graphs)
        self.flow = FlowEngine(flow_sd, dtype=dtype, device=device, use_graphs=use_graphs)
        self.dac = DacDecoderEngine(dac_sd, list(dac_rates), dtype=dtype, device=device)
        self.hop = self.dac.hop

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

    @torch.no_grad()
    def tts_batch(self, texts, flow_embeddings, seed=0, exact_steps=None, group_size=8, max_pad_ratio=1.25,
                  frame_quantum=32) -> List[torch.Tensor]:
        """Throughput path for a batch of independent utterances (BASELINE config 4, one rank's share):
        one batched AR decode for all of them, per-utterance conformer encoder, flow ODE solves batched over
        groups of similar length (zero padded + masked), DAC decode per utterance.  No prompts (synthetic load)."""
        B = len(texts)
        z = torch.zeros(1, 0, dtype=torch.long, device=self.dev)
        zf = torch.zeros(1, 0, 80, device=self.dev)
        toks = self.generate_tokens(texts, seed=seed, exact_steps=exact_steps)
        conds = [self.flow.conditions(toks[b].reshape(1, -1), z, zf, flow_embeddings[b]) for b in range(B)]
        order = sorted(range(B), key=lambda b: conds[b][0].shape[0])
        wavs: List[Optional[torch.Tensor]] = [None] * B
        i = 0
        while i < B:
            j = i + 1
            t0 = conds[order[i]][0].shape[0]
            while j < B and j - i < group_size and conds[order[j]][0].shape[0] <= max(t0 * max_pad_ratio, t0 + frame_quantum):
                j += 1
            grp = order[i:j]
            xs = self.flow.cfm_batch([conds[b][0] for b in grp], [conds[b][1] for b in grp], [conds[b][2] for b in grp],
                                     pad_to=frame_quantum)
            for b, lat in zip(grp, xs):
                T2 = lat.shape[0]
                zt = torch.empty(1, T2, 80, dtype=TORCH_DT[self.dtype], device=self.dev)
                ops.copy2d(lat, F32, 0, 80, 1, zt, self.dtype, 0, 80, 1, rows=T2, cols=80)
                wavs[b] = self.dac.decode_time_major(zt, 1, T2)
            i = j
        return wavs
