"""attn_flash_xs (the split build's estimator attention) launch time over a few flow-group shapes: a hipGraph of 56 launches, HIP
events, for every `form` of mmx_attn_flash_xs (0 = chosen per launch, 1 = 128-query workgroups, 2 = 256-query, 3 = 4-wave 64-query).

    python tools/flash_lab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
sys.path.insert(0, ROOT)
from mmx import ops  # noqa: E402
from bench import _event_time_graph  # noqa: E402

dev = "cuda"
for n, T in [(1, 500), (3, 980), (5, 420), (5, 860), (6, 330), (8, 896)]:
    B, Tp = 2 * n, ops.round_up(T, 8)
    qk = torch.randn(B, T, 2048, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, 2, 512, Tp, device=dev).to(torch.bfloat16)
    ao = torch.empty(B, T, 512, device=dev)
    fl = 4.0 * 64 * 8 * B * T * T
    for form in (0, 1, 2, 3):
        us = _event_time_graph(lambda i=0: ops.attn_flash_xs(qk, vt, ao, B=B, H=8, T=T, ldqk=2048, ldvt=Tp, ldo=512, qk_bs=T * 2048, vt_bs=2 * 512 * Tp,
                                                             o_bs=T * 512, scale=0.125, form=form), 56)
        print(f"n={n} T={T} form {form}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (algorithmic)", flush=True)
