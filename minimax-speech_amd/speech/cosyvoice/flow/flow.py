"""Drop-in for speech/cosyvoice/flow/flow.py: CausalMaskedDiffWithXvec (:201-511, inference path)."""
from typing import Dict

import torch
from torch import nn

from .. import _paths  # noqa: F401
from mmx import shapes
from mmx.shell import EngineHost, register


class CausalMaskedDiffWithXvec(EngineHost):
    def __init__(self, input_size: int = 512, output_size: int = 80, spk_embed_dim: int = 192, output_type: str = "mel",
                 vocab_size: int = 4096, input_frame_rate: int = 50, only_mask_loss: bool = True,
                 token_latent_ratio: int = 2, pre_lookahead_len: int = 3, use_speaker_encoder: bool = False,
                 freeze_speaker_encoder: bool = False, max_conditioning_inputs: int = 2,
                 speaker_encoder_path: str = None, encoder: nn.Module = None, decoder: nn.Module = None,
                 decoder_conf: Dict = None, mel_feat_conf: Dict = None):
        super().__init__()
        self.input_size, self.output_size, self.vocab_size = input_size, output_size, vocab_size
        self.output_type, self.input_frame_rate = output_type, input_frame_rate
        self.spk_embed_dim = spk_embed_dim
        self.token_latent_ratio = token_latent_ratio
        # cli/model.py:296 reads flow.token_mel_ratio, which this fork's flow.py never sets: define it
        self.token_mel_ratio = token_latent_ratio
        self.pre_lookahead_len = pre_lookahead_len
        self.use_speaker_encoder = use_speaker_encoder
        self.only_mask_loss = only_mask_loss
        man = shapes.flow_manifest(vocab=vocab_size, input_size=input_size, output_size=output_size, spk_embed_dim=spk_embed_dim)
        register(self, {k: v for k, v in man.items() if not k.startswith(("encoder.", "decoder."))})
        if use_speaker_encoder:
            from cosyvoice.llm.llm import LearnableSpeakerEncoder
            self.speaker_encoder = LearnableSpeakerEncoder(mel_dim=80, model_dim=512, output_dim=spk_embed_dim,
                                                           num_blocks=6, num_heads=8)
        self.freeze_speaker_encoder = freeze_speaker_encoder
        self.encoder = encoder
        self.decoder = decoder

    def _eng(self):
        from mmx.flow import FlowEngine
        dev = self._device()
        if self._engine is None:
            est = self.decoder.estimator
            self._engine = FlowEngine(self.state_dict(), dtype=self.compute_dtype, device=dev, wplanes=getattr(self, 'weight_planes', False),
                                      enc_chunk=self.encoder.static_chunk_size, est_chunk=est.static_chunk_size,
                                      pre_lookahead_len=self.pre_lookahead_len, cfg_rate=self.decoder.inference_cfg_rate)
            self._engine.set_noise(self.decoder.rand_noise)
        return self._engine

    @torch.inference_mode()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding=None,
                  reference_mels=None, reference_mel_lengths=None, reference_mel_masks=None, streaming=False,
                  finalize=False):
        """flow.py:437-511 -> (feat [1, 80, T2] float32, None)."""
        assert token.shape[0] == 1
        if embedding is None:
            embedding = torch.zeros(1, self.spk_embed_dim, device=token.device)
        ref = reference_mels if self.use_speaker_encoder else None
        return self._eng().inference(token, prompt_token, prompt_feat, embedding, streaming, finalize, ref), None
