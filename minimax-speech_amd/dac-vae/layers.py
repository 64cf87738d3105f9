"""Drop-in for dac-vae/layers.py (the pieces the decoder uses): snake (:18-24), Snake1d (:27-33).
Host-side statements of the ops; the decoder engine fuses Snake into the conv epilogues (csrc/gemm.hip)."""
import torch
import torch.nn as nn


def snake(x, alpha):
    shape = x.shape
    x = x.reshape(shape[0], shape[1], -1)
    x = x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)
    return x.reshape(shape)


class Snake1d(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def forward(self, x):
        return snake(x, self.alpha)
