"""Drop-in for the CosyVoice2Model glue of speech/cosyvoice/cli/model.py (:240-386): `tts` (non-streaming and the
streaming hop schedule) and `token2wav`, with the HiFT vocoder call (model.py:316) replaced by DACVAE.decode — the
composition the README and the flow's training target imply.

Streaming follows the reference's schedule (poll until token_hop_len + pre_lookahead tokens are available, re-run the
flow over all tokens so far with chunk-causal masks, emit the new part; closing pass without masks).  The HiFT source / mel
caches (model.py:258-261,298-311) have no DAC equivalent: the DAC decoder's receptive field is finite (16 latent frames to
the left, 15 to the right for configx2.yml), so each pass decodes a window with that much context and renders only the frames
whose right context is final.  Where two streaming passes meet they agree, and the chunks equal the offline decode of the
same latents; at the closing seam (the closing pass re-solves every frame) the last mel_cache_len frames are rendered by
both passes and cross-faded with fade_in_out, as the reference does at every seam (model.py:304-311).  The rule is restated
on the CPU in oracle/stream.py and checked against it in tests/test_gpu_stream.py.
"""
import threading
import time
import uuid
from typing import Generator

import numpy as np
import torch

from cosyvoice.utils.common import fade_in_out


class CosyVoice2Model:
    def __init__(self, llm: torch.nn.Module, flow: torch.nn.Module, hift: torch.nn.Module, fp16: bool = False):
        """`hift` is the waveform decoder: a dac-vae `DACVAE` in this fork (kept under the reference's argument name)."""
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.llm, self.flow, self.hift = llm, flow, hift
        # cli/model.py:250-253: fp16=True halves llm and flow, fp16=False (the default) computes in fp32.  Here: fp16=True = the
        # bf16 speed build of those two modules; fp16=False = the split build (fp32-grade: the CPU path's token ids, waveform
        # within 1e-3; weight planes when the loaded checkpoint is not bf16-representable).  A module whose build was chosen
        # explicitly (float_parity() / split_parity()) keeps it; the waveform decoder follows the flow's build.
        self.fp16 = fp16
        for m in (llm, flow, hift):
            if hasattr(m, "split_parity") and not any(getattr(x, "dtype_chosen", False) for x in m.modules()):
                m.split_parity(not fp16, chosen=False)
        self.token_hop_len = 25                           # must match the training static_chunk_size
        self.mel_cache_len = 8                            # model.py:258: frames two passes overlap by
        self.hop = int(np.prod(getattr(hift, "decoder_rates", [5, 4, 4, 3, 2])))   # samples per latent frame
        from mmx.dac import DacDecoderEngine
        self.dac_ctx_left, self.dac_ctx_right = DacDecoderEngine.receptive_field(getattr(hift, "decoder_rates", [5, 4, 4, 3, 2]))
        # model.py:262 speech_window = np.hamming(2 * source_cache_len), here over mel_cache_len DAC frames and with each pair
        # of halves scaled to sum to one (a cross-fade of two equal renderings is then the identity)
        self.speech_window = self._window(self.mel_cache_len)
        self.llm_stream = None                            # the llm_job thread launches on its own stream (mmx/flow.py rule)
        self.lock = threading.Lock()
        self.tts_speech_token_dict, self.llm_end_dict, self.hift_cache_dict = {}, {}, {}

    def _window(self, frames):
        w = np.hamming(2 * frames * self.hop)
        n = w.shape[0] // 2
        return torch.from_numpy(np.concatenate([w[:n] / (w[:n] + w[n:]), w[n:] / (w[:n] + w[n:])])).float()

    def load(self, llm_model, flow_model, hift_model):
        self.llm.load_state_dict(torch.load(llm_model, map_location="cpu"), strict=True)
        self.llm.to(self.device).eval()
        self.flow.load_state_dict(torch.load(flow_model, map_location="cpu"), strict=True)
        self.flow.to(self.device).eval()
        sd = torch.load(hift_model, map_location="cpu")
        self.hift.load_state_dict(sd.get("generator", sd), strict=True)
        self.hift.to(self.device).eval()

    def llm_job(self, text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid_):
        i32 = lambda t: torch.tensor([t.shape[1]], dtype=torch.int32, device=self.device)
        common = dict(prompt_text=prompt_text.to(self.device), prompt_text_len=i32(prompt_text),
                      prompt_speech_token=llm_prompt_speech_token.to(self.device),
                      prompt_speech_token_len=i32(llm_prompt_speech_token), embedding=llm_embedding.to(self.device))
        if self.llm_stream is None:
            self.llm_stream = torch.cuda.Stream(device=self.device)
        try:
            with torch.cuda.stream(self.llm_stream):
                if isinstance(text, Generator):           # streaming input text (cli/model.py:105-112)
                    gen = self.llm.inference_bistream(text=(t.to(self.device) for t in text), **common)
                else:
                    gen = self.llm.inference(text=text.to(self.device), text_len=i32(text), uuid=uuid_, **common)
                for tok in gen:
                    self.tts_speech_token_dict[uuid_].append(tok)
        except BaseException as e:                        # surface LM errors in tts() instead of spinning forever
            self.llm_error_dict[uuid_] = e
        finally:
            self.llm_end_dict[uuid_] = True

    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False,
                  speed=1.0):
        i32 = lambda n: torch.tensor([n], dtype=torch.int32, device=self.device)
        lat, _ = self.flow.inference(token=token.to(self.device), token_len=i32(token.shape[1]),
                                     prompt_token=prompt_token.to(self.device), prompt_token_len=i32(prompt_token.shape[1]),
                                     prompt_feat=prompt_feat.to(self.device), prompt_feat_len=i32(prompt_feat.shape[1]),
                                     embedding=embedding.to(self.device), streaming=stream, finalize=finalize)
        st = self.hift_cache_dict.get(uuid)
        start = token_offset * self.flow.token_mel_ratio
        if speed != 1.0:                                   # cli/model.py:312-314: linear resampling of the latent frames in time
            assert st is None and finalize, "speed change only support non-stream inference mode"
            from mmx import ops
            return self.hift.decode(ops.resample_linear(lat[:, :, start:].float(), int((lat.shape[2] - start) / speed)))[:, 0]
        if st is None and finalize:                        # one-shot synthesis: decode everything
            return self.hift.decode(lat[:, :, start:])[:, 0]
        # Streaming session (rule stated in mmx/pipeline.py::tts_stream and oracle/stream.py): `emitted` latent frames are
        # rendered; a pass renders the frames whose right context is final from a window whose left context is the tail of
        # ALREADY RENDERED latents.  Like the reference (model.py:306-311) a streaming pass holds the samples of its last
        # mel_cache_len frames back; the next streaming pass emits them unchanged (the passes agree), the closing pass
        # (no chunk masks: its version of every frame differs) renders them again and cross-fades (fade_in_out, :304-311).
        MC, CL, CR = self.mel_cache_len, self.dac_ctx_left, self.dac_ctx_right
        emitted, tail, held = (0, None, None) if st is None else (st["emitted"], st["tail"], st["held"])
        T2 = lat.shape[2]
        hi = T2 if finalize else T2 - CR
        if hi <= emitted:
            return lat.new_zeros(1, 0)
        re = held.shape[1] // self.hop if (finalize and held is not None) else 0     # frames the last streaming pass held back
        first = emitted - re
        ctx = None if tail is None else tail[:, :, :tail.shape[2] - re][:, :, -CL:]
        seg = lat[:, :, first:] if ctx is None or ctx.shape[2] == 0 else torch.cat([ctx, lat[:, :, first:]], dim=2)
        nctx = 0 if ctx is None else ctx.shape[2]
        wav = self.hift.decode(seg)[:, 0, nctx * self.hop:(nctx + hi - first) * self.hop]
        if finalize:
            if re:
                wav = fade_in_out(wav, held, self.speech_window if re == MC else self._window(re).to(wav.device))
            return wav
        keep = min(MC, hi - emitted)
        out = wav[:, :wav.shape[1] - keep * self.hop]
        if held is not None:
            out = torch.cat([held, out], dim=1)
        new = lat[:, :, emitted:hi]
        tail = (new if tail is None else torch.cat([tail, new], dim=2))[:, :, -(CL + MC):].clone()
        self.hift_cache_dict[uuid] = {"emitted": hi, "tail": tail, "held": wav[:, wav.shape[1] - keep * self.hop:].clone()}
        return out

    def tts(self, text=torch.zeros(1, 0, dtype=torch.int32), flow_embedding=torch.zeros(0, 192),
            llm_embedding=torch.zeros(0, 192), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
            llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), prompt_speech_feat=torch.zeros(1, 0, 80),
            source_speech_token=torch.zeros(1, 0, dtype=torch.int32), stream=False, speed=1.0, **kwargs) -> Generator:
        this_uuid = str(uuid.uuid1())
        with self.lock:
            self.tts_speech_token_dict[this_uuid], self.llm_end_dict[this_uuid] = [], False
            self.hift_cache_dict[this_uuid] = None
        if not hasattr(self, "llm_error_dict"):
            self.llm_error_dict = {}
        if source_speech_token.shape[1] == 0:
            p = threading.Thread(target=self.llm_job, args=(text, prompt_text, llm_prompt_speech_token, llm_embedding, this_uuid))
        else:
            def vc():
                self.tts_speech_token_dict[this_uuid] = source_speech_token.flatten().tolist()
                self.llm_end_dict[this_uuid] = True
            p = threading.Thread(target=vc)
        p.start()
        toks = self.tts_speech_token_dict[this_uuid]
        L = self.flow.pre_lookahead_len
        if stream:
            token_offset = 0
            pad = int(np.ceil(flow_prompt_speech_token.shape[1] / self.token_hop_len) * self.token_hop_len
                      - flow_prompt_speech_token.shape[1])
            while True:
                time.sleep(0.01)
                hop = self.token_hop_len + pad if token_offset == 0 else self.token_hop_len
                if len(toks) - token_offset >= hop + L:
                    t = torch.tensor(toks[:token_offset + hop + L]).unsqueeze(0)
                    wav = self.token2wav(t, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, token_offset,
                                         this_uuid, stream=True, finalize=False)
                    token_offset += hop
                    if wav.shape[1]:
                        yield {"tts_speech": wav.cpu()}
                if self.llm_end_dict[this_uuid] and len(toks) - token_offset < hop + L:
                    break
            p.join()
            if this_uuid in self.llm_error_dict:
                raise self.llm_error_dict.pop(this_uuid)
            t = torch.tensor(toks).unsqueeze(0)
            wav = self.token2wav(t, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, token_offset, this_uuid,
                                 finalize=True)
            yield {"tts_speech": wav.cpu()}
        else:
            p.join()
            if this_uuid in self.llm_error_dict:
                raise self.llm_error_dict.pop(this_uuid)
            t = torch.tensor(toks).unsqueeze(0)
            wav = self.token2wav(t, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, 0, this_uuid, finalize=True,
                                 speed=speed)
            yield {"tts_speech": wav.cpu()}
        with self.lock:
            self.tts_speech_token_dict.pop(this_uuid)
            self.llm_end_dict.pop(this_uuid)
            self.hift_cache_dict.pop(this_uuid)
