"""Drop-in subset of speech/matcha/models/components/decoder.py used by the CosyVoice2 estimator."""
import math

import torch


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
