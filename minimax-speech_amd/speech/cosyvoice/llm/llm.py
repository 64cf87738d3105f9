"""Drop-in for speech/cosyvoice/llm/llm.py: Qwen2Encoder (:343-371) and Qwen2LM (:374-860, inference path).

Same constructor arguments, attribute paths (`llm.model.model.embed_tokens`, `speech_embedding`, `llm_decoder`,
`stop_token_ids` ...), state-dict keys and generator semantics; the arithmetic runs in mmx.llm.LlmEngine
(hand-written HIP kernels, paged KV, device-resident sampler, hipGraph decode step).
"""
import json
import os
from typing import Callable, Generator, List, Optional

import torch
from torch import nn

from .. import _paths  # noqa: F401
from mmx import shapes, ops
from mmx._lib import F32
from mmx.shell import EngineHost, Node, register

_QWEN_DEFAULT = dict(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=24,
                     num_attention_heads=14, num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6)


def _qwen_cfg(pretrain_path):
    """`pretrain_path`: a HF checkpoint directory (config.json is read; weights come via load_state_dict like
    cli/model.py:67-75 does), a dict of Qwen2Config fields, or ''/None for the CosyVoice-BlankEN shape."""
    cfg = dict(_QWEN_DEFAULT)
    if isinstance(pretrain_path, dict):
        cfg.update(pretrain_path)
    elif pretrain_path and os.path.exists(os.path.join(pretrain_path, "config.json")):
        with open(os.path.join(pretrain_path, "config.json")) as f:
            j = json.load(f)
        cfg.update({k: j[k] for k in cfg if k in j})
        if "rope_parameters" in j and "rope_theta" in j["rope_parameters"]:
            cfg["rope_theta"] = j["rope_parameters"]["rope_theta"]
    return cfg


class LearnableSpeakerEncoder(EngineHost):
    """llm.py:34-96 (Tortoise-style conditioning encoder): mel [B,80,T] -> L2-normalised [B,output_dim]."""

    def __init__(self, mel_dim: int = 80, model_dim: int = 512, output_dim: int = 192, num_blocks: int = 6,
                 num_heads: int = 8, dropout: float = 0.0, mean_pooling: bool = False):
        super().__init__()
        if mean_pooling or model_dim // num_heads != 64:
            raise NotImplementedError("first-frame pooling and 64-d heads (the reference's instantiation) only")
        self.dim, self.mean_pooling, self.num_heads = model_dim, mean_pooling, num_heads
        register(self, shapes.speaker_encoder_manifest("speaker_encoder", mel_dim, model_dim, output_dim, num_blocks),
                 prefix="speaker_encoder.")

    def _eng(self):
        from mmx.spk import SpeakerEncoderEngine
        dev = self._device()
        if self._engine is None:
            sd = {"speaker_encoder." + k: v for k, v in self.state_dict().items()}
            self._engine = SpeakerEncoderEngine(sd, dtype=self.compute_dtype, device=dev, heads=self.num_heads)
        return self._engine

    @torch.inference_mode()
    def forward(self, x, mask=None):
        return self._eng().encode(x)


class Qwen2Encoder(nn.Module):
    def __init__(self, pretrain_path):
        super().__init__()
        self.cfg = c = _qwen_cfg(pretrain_path)
        man = shapes.llm_manifest(vocab=c["vocab_size"], hidden=c["hidden_size"], inter=c["intermediate_size"],
                                  layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                                  kv_heads=c["num_key_value_heads"],
                                  head_dim=c["hidden_size"] // c["num_attention_heads"])
        register(self, man, prefix="llm.")                     # -> self.model.model.{embed_tokens,layers,norm}
        et = self.model.model.embed_tokens
        et.forward = lambda ids: torch.nn.functional.embedding(ids, et.weight)   # callers use it as a module (llm.py:694)
        self.compute_dtype, self.max_ctx, self._engine = 1, 2048, None

    class _Cache:
        """Stands for HF's past_key_values: the KV rows live in the engine's paged cache, this records how many."""

        def __init__(self, rows):
            self.rows = rows

        def get_seq_length(self):
            return self.rows

    def _eng(self):
        from mmx.llm import LlmEngine
        p = next(self.parameters())
        if not p.is_cuda:
            raise RuntimeError("Qwen2Encoder: the MI355X hot path has no CPU fallback; move the module to a ROCm device")
        if self._engine is None:
            c = self.cfg
            sd = {"llm." + k: v for k, v in self.state_dict().items()}
            self._engine = LlmEngine(sd, dtype=(3 if self.compute_dtype == 2 else self.compute_dtype), device=p.device, max_batch=1, max_ctx=self.max_ctx, wplanes=getattr(self, 'weight_planes', False),
                                     heads=c["num_attention_heads"], kv_heads=c["num_key_value_heads"],
                                     head_dim=c["hidden_size"] // c["num_attention_heads"], rope_theta=c["rope_theta"],
                                     eps=c["rms_norm_eps"])
        return self._engine

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    @torch.inference_mode()
    def forward_one_step(self, xs, masks, cache=None):
        """llm.py:359-371: xs [1, q, H] appended after `cache` -> (hidden_states[-1] [1, q, H], new cache).  `masks` is the
        reference's lower-triangular [1, q, q]; attention is full causal over the cache (the pinned-stack semantics,
        SURVEY.md §7 "version-drift trap"), which for a lower-triangular mask is the same thing."""
        assert xs.shape[0] == 1, "the AR loop decodes one sequence (llm.py:745-760)"
        rows = 0 if cache is None else cache.rows
        h = self._eng().forward_rows(xs[0], rows)
        return h.unsqueeze(0), Qwen2Encoder._Cache(rows + xs.shape[1])

    @torch.inference_mode()
    def forward(self, xs: torch.Tensor, xs_lens: torch.Tensor):
        """llm.py:349-357 (teacher-forced pass of one unpadded sequence): -> (hidden [1, T, H], masks [1, 1, T])."""
        assert xs.shape[0] == 1 and int(xs_lens[0]) == xs.shape[1]
        h = self._eng().forward_rows(xs[0], 0)
        return h.unsqueeze(0), torch.ones(1, 1, xs.shape[1], dtype=torch.bool, device=h.device)


class Qwen2LM(EngineHost):
    def __init__(self, llm_input_size: int, llm_output_size: int, speech_token_size: int, llm: nn.Module,
                 sampling: Callable, length_normalized_loss: bool = True, lsm_weight: float = 0.0,
                 mix_ratio: List[int] = [5, 15], use_speaker_encoder: bool = False, spk_embed_dim: int = 192,
                 max_conditioning_inputs: int = 2):
        super().__init__()
        self.llm_input_size, self.llm_output_size = llm_input_size, llm_output_size
        self.speech_token_size = speech_token_size
        self.use_speaker_encoder, self.spk_embed_dim = use_speaker_encoder, spk_embed_dim
        self.max_conditioning_inputs = max_conditioning_inputs
        self.sos_eos, self.task_id, self.fill_token = 0, 1, 2
        self.llm = llm
        c = llm.cfg
        man = shapes.llm_manifest(vocab=c["vocab_size"], hidden=c["hidden_size"], inter=c["intermediate_size"],
                                  layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                                  kv_heads=c["num_key_value_heads"], head_dim=c["hidden_size"] // c["num_attention_heads"],
                                  speech_token_size=speech_token_size, spk_embed_dim=spk_embed_dim)
        register(self, {k: v for k, v in man.items() if not k.startswith("llm.")})
        if use_speaker_encoder:
            self.speaker_encoder = LearnableSpeakerEncoder(mel_dim=80, model_dim=512, output_dim=spk_embed_dim, num_blocks=6,
                                                           num_heads=8)
        self.sampling = sampling
        self.mix_ratio = mix_ratio
        self.stop_token_ids = [speech_token_size + i for i in range(3)]
        self.seed = 0
        self.max_ctx = 2048

    def _sampling_kwargs(self):
        kw = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1)
        kw.update(getattr(self.sampling, "keywords", None) or {})
        return kw

    def engine(self, max_batch=1):
        from mmx.llm import LlmEngine
        dev = self._device()
        if self._engine is None or self._engine.B != max_batch:
            c = self.llm.cfg
            self._engine = LlmEngine(self.state_dict(), dtype=(3 if self.compute_dtype == 2 else self.compute_dtype), device=dev, max_batch=max_batch, wplanes=getattr(self, 'weight_planes', False),
                                     max_ctx=self.max_ctx, heads=c["num_attention_heads"], kv_heads=c["num_key_value_heads"],
                                     head_dim=c["hidden_size"] // c["num_attention_heads"], rope_theta=c["rope_theta"],
                                     eps=c["rms_norm_eps"], speech_token_size=self.speech_token_size)
            kw = self._sampling_kwargs()
            self._engine.top_p, self._engine.top_k = kw["top_p"], kw["top_k"]
            self._engine.win_size, self._engine.tau_r = kw["win_size"], kw["tau_r"]
        return self._engine

    @torch.inference_mode()
    def inference_spk(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len,
                      embedding=None, reference_mels=None, reference_mel_lengths=None, reference_mel_masks=None,
                      sampling: int = 25, max_token_text_ratio: float = 20, min_token_text_ratio: float = 2,
                      uuid: str = "") -> Generator[int, None, None]:
        """llm.py:616-674: like `inference` with a speaker-conditioning row after <sos>: from the learnable speaker
        encoder on `reference_mels` [1,N,80,T], else from `embedding` [1,192], else zeros."""
        eng = self.engine(1)
        tl = int(text.shape[1])
        text_len += prompt_text_len
        if self.use_speaker_encoder and reference_mels is not None:
            e = self.speaker_encoder._eng().reference_embedding(reference_mels)
        elif embedding is not None and embedding.shape[0] != 0:
            e = embedding
        else:
            e = None
        if e is not None:
            from mmx import ops as _ops
            if getattr(self, "_spk_w", None) is None or self._engine_spk is not eng:
                self._spk_w = _ops.pack_linear(self.spk_embed_affine_layer.weight.detach().float(), eng.dtype)
                self._spk_b = self.spk_embed_affine_layer.bias.detach().float().contiguous()
                self._engine_spk = eng
            spk = eng.speaker_conditioning(self._spk_w, self._spk_b, e)
        else:
            spk = torch.zeros(1, self.llm_input_size, device=eng.dev)
        x = eng.build_lm_input(text, prompt_text, prompt_speech_token, speaker_embed=spk)
        yield from self._decode_loop(eng, x, int(tl * min_token_text_ratio), int(tl * max_token_text_ratio))

    @torch.inference_mode()
    def inference(self, text: torch.Tensor, text_len: torch.Tensor, prompt_text: torch.Tensor,
                  prompt_text_len: torch.Tensor, prompt_speech_token: torch.Tensor,
                  prompt_speech_token_len: torch.Tensor, embedding: torch.Tensor, sampling: int = 25,
                  max_token_text_ratio: float = 20, min_token_text_ratio: float = 2,
                  uuid: str = "") -> Generator[int, None, None]:
        """llm.py:676-711 + :745-760.  Yields python ints.  (`embedding` is unused by the reference here too.)"""
        eng = self.engine(1)
        tl = int(text.shape[1])
        text_len += prompt_text_len                      # the reference mutates text_len in place (llm.py:693)
        x = eng.build_lm_input(text, prompt_text, prompt_speech_token)
        yield from self._decode_loop(eng, x, int(tl * min_token_text_ratio), int(tl * max_token_text_ratio))

    @torch.inference_mode()
    def inference_wrapper(self, lm_input, sampling, min_len, max_len, uuid):
        """llm.py:713-760, non-vLLM branch: the AR loop over a prepared lm_input [1, L, H] — batched prompt pass, captured
        decode step, log-softmax + RAS + stop / skip rules on the device (mmx/llm.py, csrc/sampler.hip)."""
        yield from self._decode_loop(self.engine(1), lm_input[0], int(min_len), int(max_len))

    def _is_device_sampler(self):
        from cosyvoice.utils.common import ras_sampling
        return getattr(self.sampling, "func", self.sampling) is ras_sampling

    def sampling_ids(self, weighted_scores: torch.Tensor, decoded_tokens: List, sampling: int, ignore_eos: bool = True):
        """llm.py:259-274 around a host `sampling` callable (used by inference_bistream when `sampling` is not RAS)."""
        num_trials, max_trials = 0, 100
        while True:
            top_ids = self.sampling(weighted_scores, decoded_tokens, sampling)
            if (not ignore_eos) or (self.speech_token_size not in top_ids):
                break
            num_trials += 1
            if num_trials > max_trials:
                raise RuntimeError("sampling reaches max_trials {} and still get eos when ignore_eos is True, check your "
                                   "input!".format(max_trials))
        return top_ids

    @torch.inference_mode()
    def inference_bistream(self, text: Generator, prompt_text: torch.Tensor, prompt_text_len: torch.Tensor,
                           prompt_speech_token: torch.Tensor, prompt_speech_token_len: torch.Tensor,
                           embedding: torch.Tensor, sampling: int = 25, max_token_text_ratio: float = 20,
                           min_token_text_ratio: float = 2) -> Generator[int, None, None]:
        """llm.py:762-870: text arrives as a generator of id tensors and is interleaved with speech tokens
        mix_ratio[0] : mix_ratio[1]; the fill token (speech_token_size + 2) asks for the next text block.  The
        interleaving loop is host logic (it waits on the caller's generator); every LM pass, log-softmax and the RAS
        draw run on the device.  A `sampling` callable other than ras_sampling is honoured on the host from the
        device log-probs, as the reference would call it."""
        eng = self.engine(1)
        dev_sampler = self._is_device_sampler()
        eng.open_stream(seed=self.seed, want_logp=not dev_sampler)
        fill, n_speech = self.speech_token_size + 2, self.speech_token_size
        mix = self.mix_ratio

        def step(x, ignore_eos, out_tokens, forced=None):
            tok = eng.feed(x, ignore_eos)
            if forced is not None:                        # llm.py:824-826: the pass runs, its draw is not used
                return forced
            if not dev_sampler:
                tok = int(self.sampling_ids(eng.logp[0], out_tokens, sampling, ignore_eos=ignore_eos))
            return tok

        sos = eng.llm_emb[0:1]
        task = eng.llm_emb[1:2]
        pse = eng.embed_speech(prompt_speech_token) if int(prompt_speech_token_len) != 0 else torch.zeros(0, eng.H, device=eng.dev)
        lm_input = sos
        pending = False                                   # lm_input is exactly the last accepted token's embedding
        out_tokens: List[int] = []
        text_cache = eng.embed_text(prompt_text)
        next_fill_index = -1
        for this_text in text:
            text_cache = torch.cat([text_cache, eng.embed_text(this_text)], dim=0)
            while pse.shape[0] != 0:
                if text_cache.shape[0] >= mix[0]:
                    lm_input = torch.cat([lm_input, text_cache[:mix[0]], pse[:mix[1]]], dim=0)
                    pending = False
                    text_cache, pse = text_cache[mix[0]:], pse[mix[1]:]
                else:
                    break
            if pse.shape[0] == 0:
                if (len(out_tokens) != 0 and out_tokens[-1] == fill) or (len(out_tokens) == 0 and lm_input.shape[0] == 1):
                    if text_cache.shape[0] >= mix[0]:
                        lm_input_text = text_cache[:mix[0]]
                        if len(out_tokens) != 0 and out_tokens[-1] == fill:
                            lm_input = lm_input_text
                        else:
                            lm_input = torch.cat([lm_input, lm_input_text], dim=0)
                        pending = False
                        text_cache = text_cache[mix[0]:]
                    else:
                        continue
                while True:
                    force = next_fill_index != -1 and len(out_tokens) == next_fill_index
                    top_ids = step(None if pending else lm_input, True, out_tokens, fill if force else None)
                    if force:
                        next_fill_index += mix[1] + 1
                    if top_ids == fill:
                        next_fill_index = len(out_tokens) + mix[1] + 1
                    out_tokens.append(top_ids)
                    eng.commit(top_ids)
                    if top_ids >= n_speech:
                        if top_ids == fill:
                            break
                        raise ValueError("should not get token {}".format(top_ids))
                    yield top_ids
                    lm_input, pending = eng.x_in[0:1], True
        # final decode: the remaining text, then <task_id>, until eos (llm.py:848-870)
        lm_input = torch.cat([lm_input.clone(), text_cache, task], dim=0)
        pending = False
        while True:
            top_ids = step(None if pending else lm_input, False, out_tokens)
            out_tokens.append(top_ids)
            eng.commit(top_ids)
            if top_ids >= n_speech:
                if top_ids == n_speech:
                    break
                raise ValueError("should not get token {}".format(top_ids))
            yield top_ids
            lm_input, pending = eng.x_in[0:1], True

    def _decode_loop(self, eng, x, min_len, max_len):
        eng.start([x], [min_len], [max_len], seed=self.seed)
        sent = 0
        done = 1
        while True:
            st = eng.state[:, 0].tolist()
            toks = eng.out_tokens[0, sent:st[2]].tolist()
            for t in toks:
                yield t
            sent = st[2]
            if st[3] or done >= max_len:
                break
            for _ in range(min(8, max_len - done)):
                eng.step()
                done += 1


class TransformerLM(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("CosyVoice-1 TransformerLM (llm.py:99-340) is not instantiated by config.yaml: out of scope")
