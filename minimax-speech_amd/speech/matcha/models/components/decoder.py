"""Drop-in subset of speech/matcha/models/components/decoder.py used by the CosyVoice2 estimator:
SinusoidalPosEmb (:14-29), Block1D (:32-43), ResnetBlock1D (:46-61), TimestepEmbedding (:73-117).

Same constructor arguments and state-dict keys (torch containers hold the parameters and are never called); `forward`
runs the block on the HIP kernels (mmx/blocks.py).  The estimator itself (cosyvoice.flow.decoder) executes the same
arithmetic in the fused row-tile kernels; these classes make the block-level API a working drop-in and give every
sub-block its own parity test (tests/test_gpu_blocks.py)."""
import math

import torch
from torch import nn

from ... import _paths  # noqa: F401
from mmx.shell import EngineHost


class SinusoidalPosEmb(torch.nn.Module):
    """decoder.py:14-29 (host-side statement; the estimator computes it in csrc/elementwise.hip)."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0, "SinusoidalPosEmb requires dim to be even"
        self.dim = dim

    def forward(self, x, scale=1000):
        if x.ndim < 1:
            x = x.unsqueeze(0)
        half = self.dim // 2
        e = math.log(10000) / (half - 1)
        e = torch.exp(torch.arange(half, device=x.device).float() * -e)
        e = scale * x.unsqueeze(1) * e.unsqueeze(0)
        return torch.cat((e.sin(), e.cos()), dim=-1)


class BlockHost(EngineHost):
    """EngineHost whose engine is mmx.blocks.BlockOps over the module's own state dict."""

    def _ops(self):
        from mmx.blocks import BlockOps
        self._device()
        if self._engine is None:
            self._engine = BlockOps(self.state_dict(), self.compute_dtype)
        return self._engine


class Block1D(BlockHost):
    causal = False

    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.groups = groups
        self.block = nn.Sequential(nn.Conv1d(dim, dim_out, 3, padding=1), nn.GroupNorm(groups, dim_out), nn.Mish())

    @torch.inference_mode()
    def forward(self, x, mask):
        """x [B, dim, T], mask [B, 1, T] -> conv3(x * mask) -> norm -> Mish, * mask   (decoder.py:41-43)."""
        return self._ops().block1d(x, mask, self.causal, self.groups)


class ResnetBlock1D(BlockHost):
    causal = False

    def __init__(self, dim, dim_out, time_emb_dim, groups=8):
        super().__init__()
        self.groups = groups
        self.mlp = nn.Sequential(nn.Mish(), nn.Linear(time_emb_dim, dim_out))
        self.block1 = Block1D(dim, dim_out, groups=groups)
        self.block2 = Block1D(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv1d(dim, dim_out, 1)

    @torch.inference_mode()
    def forward(self, x, mask, time_emb):
        """decoder.py:56-61: block1 -> + mlp(time_emb) -> block2 -> + res_conv(x * mask)."""
        return self._ops().resnet(x, mask, time_emb, self.causal, self.groups)


class TimestepEmbedding(BlockHost):
    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu", out_dim: int = None,
                 post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        if act_fn != "silu" or post_act_fn is not None or cond_proj_dim is not None:
            raise NotImplementedError("the estimator's instantiation only: act_fn='silu' (flow/decoder.py:325-329)")
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    @torch.inference_mode()
    def forward(self, sample, condition=None):
        assert condition is None
        return self._ops().timestep_embedding(sample)
