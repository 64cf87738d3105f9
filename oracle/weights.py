"""oracle/weights.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Deterministic synthetic weights.  No pretrained weights exist offline, and the
full-size state dicts (35 M / 113 M / 494 M parameters) cannot be committed as fixtures.  Instead every
tensor is a pure function of (its state-dict key, its shape, a seed) on torch's CPU generator, so

  * oracle/gen_golden.py loads these weights INTO the reference's own modules, runs them, and commits
    only inputs + outputs (+ the key->shape manifest) under tests/golden/;
  * tests and bench regenerate the identical weights anywhere (same torch build) without the reference.

The scales are chosen so activations stay O(1) through the deep stacks (56 transformer blocks, 37
convs): random-init reference weights give a ~1e-3 waveform, which would make the 1e-3 abs parity
tolerance of BASELINE.json vacuous.
"""
import json
import os
import sys
from typing import Dict, Tuple

import torch

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "minimax-speech_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from mmx.synth import synth_state_dict, synth_tensor  # noqa: E402,F401  (one generator shared with bench.py)


def load_manifest(path: str) -> Dict[str, Tuple[int, ...]]:
    with open(path) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def save_manifest(sd, path: str):
    with open(path, "w") as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)


def sampler_cases(n: int = 200, seed: int = 31):
    """Inputs of the sampler golden cases (tests/golden/sampler.npz holds the reference's outputs; the
    torch seed of case s is 1000 + s): log-prob vectors of four peakedness levels and decoded-token
    histories of length 0..12, every third one ending in the arg-max token (forces the repetition branch)."""
    g = torch.Generator().manual_seed(seed)
    cases = []
    for s in range(n):
        scale = [0.5, 2.0, 4.0, 8.0][s % 4]
        logp = (torch.randn(6564, generator=g) * scale).log_softmax(0)
        hist = torch.randint(0, 6561, (s % 13,), generator=g).tolist()
        if s % 3 == 0 and hist:
            hist[-1] = int(logp.argmax())
        cases.append((logp, hist))
    return cases
