"""est_tail_kernel launch time by tile height / waves / ring depth over a few group shapes (a hipGraph of 112 launches rotating
over the 56 mid blocks' weights, HIP events) — the numbers behind FlowEngine._WG_US and the tiling rule.

    python tools/tail_lab.py [bf16|x]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
sys.path.insert(0, ROOT)
from mmx import ops, shapes, synth  # noqa: E402
from mmx.flow import FlowEngine  # noqa: E402
from bench import _event_time_graph  # noqa: E402

dt = {"bf16": 1, "x": 2}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
fl = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=dt, use_graphs=False)
blocks = [w for st in fl.mid for w in st["blocks"]]
split = dt == 2


def run(n, T, bm, waves, pf):
    B, C = 2 * n, fl.C
    Tp = ops.round_up(T, 8)
    ao = torch.randn(B, T, 512, device=fl.dev).to(fl.tdt)
    x = torch.randn(B, T, C, device=fl.dev)
    if split:
        qk = torch.empty(B, T, 2048, dtype=torch.bfloat16, device=fl.dev)
        vt = torch.zeros(B, 2, 512, Tp, dtype=torch.bfloat16, device=fl.dev)
        ldq, vt_bs = 2048, 2 * 512 * Tp
    else:
        qk, vt = fl._new(B, T, 1024), torch.zeros(B, 512, Tp, dtype=fl.tdt, device=fl.dev)
        ldq, vt_bs = 1024, 512 * Tp

    def one(i=0):
        w, wn = blocks[i % len(blocks)], blocks[(i + 1) % len(blocks)]
        nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp,
                           vt_bs=vt_bs)
        ops.est_tail(ao, x, w, B=B, T=T, dtype=dt, bm=bm, nxt=nxt, waves=waves, pf=pf)

    return _event_time_graph(one, 2 * len(blocks))


cfgs = ([(32, 8, 0), (32, 4, 0), (16, 8, 0), (16, 4, 0)] if split else
        [(64, 4, 2), (64, 4, 4), (64, 8, 2), (32, 8, 2), (32, 8, 4), (32, 4, 4), (32, 4, 2), (16, 8, 4), (16, 4, 8)])
for n, T in [(1, 500), (2, 1000), (4, 1000), (5, 1000), (6, 1000), (8, 896), (8, 1000), (12, 1000), (16, 1000)]:
    rows = 2 * n * T
    line = []
    for bm, waves, pf in cfgs:
        try:
            us = run(n, T, bm, waves, pf)
            line.append(f"{bm}x{waves}w/pf{pf}: {us:6.1f} ({-(-T // bm) * 2 * n:4d} wg)")
        except Exception as e:  # noqa: BLE001
            line.append(f"{bm}x{waves}w/pf{pf}: {type(e).__name__}")
    print(f"rows {rows:6d} | " + " | ".join(line), flush=True)
