"""Parameter shells: nn.Modules whose parameters are registered from a state-dict manifest (mmx/shapes.py), so
the drop-in classes expose the reference's state_dict()/load_state_dict()/attribute paths while
all arithmetic runs in the HIP engines (mmx/{llm,flow,dac}.py)."""
from typing import Dict, Tuple

import torch
from torch import nn


class Node(nn.Module):
    """Anonymous container (a Sequential / ModuleList / Linear stand-in that only holds parameters)."""

    def __getitem__(self, i):
        return getattr(self, str(i))


def register(root: nn.Module, manifest: Dict[str, Tuple[int, ...]], prefix: str = "", device=None):
    """Creates nested containers and parameters for every manifest key that starts with `prefix`."""
    for key, shape in manifest.items():
        if not key.startswith(prefix):
            continue
        parts = key[len(prefix):].split(".")
        m = root
        for p in parts[:-1]:
            if not hasattr(m, p):
                m.add_module(p, Node())
            m = getattr(m, p)
        m.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape, device=device), requires_grad=False))
    return root


class EngineHost(nn.Module):
    """Mixin: lazily builds (and caches) a HIP engine from the module's current state_dict; any parameter load
    or dtype/device move invalidates it."""

    _engine = None
    compute_dtype = 1          # BF16 by default; .float_parity() switches to the exact-fp32 build
    dtype_chosen = False       # True once float_parity() / split_parity() was called (CosyVoice2Model's fp16 flag picks otherwise)
    # split build: "auto" = the checkpoint's weights are carried as bf16 planes (include/mmx_hip.h MMX_X2W / MMX_X3W) exactly
    # when they are not bf16-representable, so a module loaded from the reference's fp32 llm.pt / flow.pt / DAC generator
    # (cli/model.py:67-75, dac-vae/inference.py:42-46) is not rounded at load; True / False force it
    weight_planes = "auto"

    def _invalidate(self):
        self._engine = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        for m in self.modules():
            if isinstance(m, EngineHost):
                m._invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def float_parity(self, on=True):
        """fp32 storage + exact-fp32 MFMA (the parity build) instead of bf16."""
        for m in self.modules():
            if isinstance(m, EngineHost):
                m.compute_dtype, m.dtype_chosen = (0 if on else 1), True
                m._invalidate()
        return self

    def split_parity(self, on=True, weight_planes="auto", chosen=True):
        """The split build (mmx/_lib.py X2 / X3): bf16 weight stream, fp32 activations carried as bf16 terms inside the MFMA
        products - token ids and waveform as the fp32 build's, at (nearly) the bf16 build's speed."""
        for m in self.modules():
            if isinstance(m, EngineHost):
                m.compute_dtype, m.dtype_chosen, m.weight_planes = (2 if on else 1), chosen, weight_planes
                m._invalidate()
        return self

    def _device(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise RuntimeError(f"{type(self).__name__}: the MI355X hot path has no CPU fallback; move the module to "
                               "a ROCm device (.to('cuda')) and build minimax-speech_amd/lib/libmmx_hip.so")
        return p.device
