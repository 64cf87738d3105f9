"""Drop-in for speech/cosyvoice/transformer/upsample_encoder.py: UpsampleConformerEncoder (:105-330)."""
from typing import Tuple

import torch
from torch import nn

from .. import _paths  # noqa: F401
from mmx import shapes
from mmx.shell import EngineHost, register


class UpsampleConformerEncoder(EngineHost):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.0, input_layer: str = "conv2d", pos_enc_layer_type: str = "rel_pos",
                 normalize_before: bool = True, static_chunk_size: int = 0, use_dynamic_chunk: bool = False,
                 global_cmvn: nn.Module = None, use_dynamic_left_chunk: bool = False,
                 positionwise_conv_kernel_size: int = 1, macaron_style: bool = True,
                 selfattention_layer_type: str = "rel_selfattn", activation_type: str = "swish",
                 use_cnn_module: bool = True, cnn_module_kernel: int = 15, causal: bool = False,
                 cnn_module_norm: str = "batch_norm", key_bias: bool = True, gradient_checkpointing: bool = False):
        super().__init__()
        if not (input_layer == "linear" and pos_enc_layer_type == "rel_pos_espnet" and selfattention_layer_type == "rel_selfattn"
                and not use_cnn_module and not macaron_style and normalize_before and output_size == 512 and input_size == 512):
            raise NotImplementedError("only the configuration of speech/config.yaml:73-89 is on the hot path")
        self._output_size = output_size
        self.static_chunk_size = static_chunk_size
        self.num_blocks, self.attention_heads, self.linear_units = num_blocks, attention_heads, linear_units
        man = shapes.flow_manifest(heads=attention_heads, linear_units=linear_units, num_blocks=num_blocks)
        register(self, man, prefix="encoder.")

    def output_size(self) -> int:
        return self._output_size

    def _eng(self):
        from mmx.flow import FlowEngine
        dev = self._device()
        if self._engine is None:
            sd = {"encoder." + k: v for k, v in self.state_dict().items()}
            self._engine = FlowEngine(sd, dtype=self.compute_dtype, device=dev, wplanes=getattr(self, 'weight_planes', False), enc_chunk=self.static_chunk_size,
                                      parts=("encoder",))
        return self._engine

    @torch.inference_mode()
    def forward(self, xs, xs_lens, context=torch.zeros(0, 0, 0), decoding_chunk_size: int = 0,
                num_decoding_left_chunks: int = -1, streaming: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """upsample_encoder.py:243-316: xs [1, T, 512] embedded tokens (+ `context` [1, 3, 512], the look-ahead rows of
        a non-final streaming call) -> (h [1, 2T, 512] fp32, masks [1, 1, 2T] bool).  Batch 1, as flow.inference
        calls it (flow.py:453 asserts it)."""
        assert xs.shape[0] == 1 and int(xs_lens[0]) == xs.shape[1], "the hot path encodes one unpadded utterance"
        eng = self._eng()
        has_ctx = context is not None and context.numel() > 0
        rows = torch.cat([xs[0], context[0].to(xs.device)], dim=0) if has_ctx else xs[0]
        a0 = rows.to(eng.dev, eng.tdt).contiguous()
        h = eng.encode_embedded(a0, finalize=not has_ctx, streaming=streaming, hidden=True)
        return h.unsqueeze(0), torch.ones(1, 1, h.shape[0], dtype=torch.bool, device=h.device)
