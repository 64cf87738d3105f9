"""Drop-in for speech/cosyvoice/flow/decoder.py: CausalConditionalDecoder (:294-496), the CFM estimator."""
import torch

from .. import _paths  # noqa: F401
from mmx import shapes
from mmx.shell import EngineHost, register
from matcha.models.components.decoder import Block1D, ResnetBlock1D


class CausalConv1d(torch.nn.Conv1d):
    """flow/decoder.py:36-62: Conv1d with kernel_size - 1 zeros of left padding; runs as one windowed-GEMM launch."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1, groups: int = 1,
                 bias: bool = True, padding_mode: str = "zeros", device=None, dtype=None) -> None:
        super().__init__(in_channels, out_channels, kernel_size, stride, padding=0, dilation=dilation, groups=groups, bias=bias,
                         padding_mode=padding_mode, device=device, dtype=dtype)
        assert stride == 1 and groups == 1 and dilation == 1
        self.causal_padding = kernel_size - 1
        self.compute_dtype = 1

    @torch.inference_mode()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from mmx import ops
        from mmx._lib import F32, TORCH_DT
        if not self.weight.is_cuda:
            raise RuntimeError("CausalConv1d: the MI355X hot path has no CPU fallback; move the module to a ROCm device")
        dt, dev = self.compute_dtype, self.weight.device
        B, cin, T = x.shape
        a = torch.empty(B, T, cin, dtype=TORCH_DT[dt], device=dev)
        ops.copy2d(x.to(dev, torch.float32).contiguous(), F32, cin * T, 1, T, a, dt, T * cin, cin, 1, rows=T, cols=cin, batch=B)
        w = ops.pack_conv1d(self.weight.detach().float(), dt)
        y = torch.empty(B, T, w.shape[0], dtype=torch.float32, device=dev)
        ops.conv1d(a, w, T=T, Cin=cin, k=self.kernel_size[0], pad_left=self.causal_padding, dtype=dt, batch=B,
                   bias=(self.bias.detach().float().contiguous() if self.bias is not None else None), out_f32=y)
        out = torch.empty(B, w.shape[0], T, dtype=torch.float32, device=dev)
        ops.copy2d(y, F32, T * w.shape[0], w.shape[0], 1, out, F32, w.shape[0] * T, 1, T, rows=T, cols=w.shape[0], batch=B)
        return out


class Transpose(torch.nn.Module):
    def __init__(self, dim0: int, dim1: int):
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1

    def forward(self, x):
        return torch.transpose(x, self.dim0, self.dim1)


class CausalBlock1D(Block1D):
    """flow/decoder.py:65-77: Sequential(CausalConv1d k3, Transpose, LayerNorm, Transpose, Mish) on x * mask, * mask."""
    causal = True

    def __init__(self, dim: int, dim_out: int):
        super().__init__(dim, dim_out)
        self.block = torch.nn.Sequential(CausalConv1d(dim, dim_out, 3), Transpose(1, 2), torch.nn.LayerNorm(dim_out),
                                         Transpose(1, 2), torch.nn.Mish())


class CausalResnetBlock1D(ResnetBlock1D):
    """flow/decoder.py:80-85."""
    causal = True

    def __init__(self, dim: int, dim_out: int, time_emb_dim: int, groups: int = 8):
        super().__init__(dim, dim_out, time_emb_dim, groups)
        self.block1 = CausalBlock1D(dim, dim_out)
        self.block2 = CausalBlock1D(dim_out, dim_out)


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
            self._engine = FlowEngine(sd, dtype=self.compute_dtype, device=dev, wplanes=getattr(self, 'weight_planes', False), est_chunk=self.static_chunk_size,
                                      parts=("estimator",))
        return self._engine

    @torch.inference_mode()
    def forward(self, x, mask, mu, t, spks=None, cond=None, streaming=False):
        """The estimator seam (flow_matching.py:128-131): [B,80,T] tensors in, [B,80,T] out."""
        B, _, T = x.shape
        z = torch.zeros_like(x)
        spks = spks if spks is not None else torch.zeros(B, 80, device=x.device)
        return self._eng().estimator_channels_first(x, mask, mu, t, spks, cond if cond is not None else z, streaming)
