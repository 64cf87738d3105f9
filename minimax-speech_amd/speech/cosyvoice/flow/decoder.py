"""Drop-in for speech/cosyvoice/flow/decoder.py: CausalConditionalDecoder (:294-496), the CFM estimator."""
import torch

from .. import _paths  # noqa: F401
from mmx import shapes
from mmx.shell import EngineHost, register


class CausalConditionalDecoder(EngineHost):
    def __init__(self, in_channels, out_channels, channels=(256, 256), dropout=0.05, attention_head_dim=64, n_blocks=1,
                 num_mid_blocks=2, num_heads=4, act_fn="snake", static_chunk_size=50, num_decoding_left_chunks=2):
        super().__init__()
        channels = tuple(channels)
        if not (len(channels) == 1 and act_fn == "gelu" and attention_head_dim == 64):
            raise NotImplementedError("only channels=[C], act_fn='gelu', head_dim 64 (speech/config.yaml:104-116) is built")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.static_chunk_size, self.num_decoding_left_chunks = static_chunk_size, num_decoding_left_chunks
        self.hp = dict(est_in=in_channels, est_ch=channels[0], n_blocks=n_blocks, num_mid_blocks=num_mid_blocks,
                       est_heads=num_heads, head_dim=attention_head_dim, output_size=out_channels)
        man = shapes.flow_manifest(**self.hp)
        register(self, man, prefix="decoder.estimator.")

    def _eng(self):
        from mmx.flow import FlowEngine
        dev = self._device()
        if self._engine is None:
            sd = {"decoder.estimator." + k: v for k, v in self.state_dict().items()}
            self._engine = FlowEngine(sd, dtype=self.compute_dtype, device=dev, est_chunk=self.static_chunk_size,
                                      parts=("estimator",))
        return self._engine

    @torch.inference_mode()
    def forward(self, x, mask, mu, t, spks=None, cond=None, streaming=False):
        """The estimator seam (flow_matching.py:128-131): [B,80,T] tensors in, [B,80,T] out."""
        B, _, T = x.shape
        z = torch.zeros_like(x)
        spks = spks if spks is not None else torch.zeros(B, 80, device=x.device)
        return self._eng().estimator_channels_first(x, mask, mu, t, spks, cond if cond is not None else z, streaming)
