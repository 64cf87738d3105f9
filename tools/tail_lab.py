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

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"     # xw: the split build on an fp32-kind checkpoint (weight planes)
dt = {"bf16": 1, "x": 2, "xw": 2}[mode]
fl = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0, kind=("fp32" if mode == "xw" else "bf16")), dtype=dt, use_graphs=False,
                wplanes=(mode == "xw"))
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
        ops.est_tail(ao, x, w, B=B, T=T, dtype=dt, bm=bm, nxt=nxt, waves=waves % 100, pf=pf, occ2=100 <= waves < 200, narrow=200 <= waves < 300)

    return _event_time_graph(one, 2 * len(blocks))


def sweep():
    # waves 2xx: 8 waves with 32-column passes
    cfgs = ([(32, 8, 0), (64, 208, 2), (16, 8, 0)] if mode == "xw" else [(32, 8, 0), (64, 208, 2), (64, 208, 4), (16, 8, 0), (16, 4, 0)] if split else
            [(64, 4, 2), (64, 208, 2), (32, 208, 8), (16, 8, 4)])
    for n, T in [(1, 150), (1, 500), (2, 1000), (4, 1000), (5, 1000), (6, 1000), (8, 896), (8, 1000), (12, 1000), (16, 1000)]:
        rows = 2 * n * T
        line = []
        for bm, waves, pf in cfgs:
            try:
                us = run(n, T, bm, waves, pf)
                line.append(f"{bm}x{waves}w/pf{pf}: {us:6.1f} ({-(-T // bm) * 2 * n:4d} wg)")
            except Exception as e:  # noqa: BLE001
                line.append(f"{bm}x{waves}w/pf{pf}: {type(e).__name__}")
        print(f"rows {rows:6d} | " + " | ".join(line), flush=True)


def stamps(n=5, T=1000, bm=64, waves=4, pf=2):
    """Stage-boundary shader-clock stamps of every wave of one launch (mmx_lab_tail_stamps): median over workgroups of
    the time between consecutive stamps of wave 0..NW-1, in us (s_memtime ticks / 2100)."""
    import ctypes as C
    from mmx import _lib
    lib = _lib.load()
    B = 2 * n
    nwg = -(-T // bm) * B
    nw = (waves % 100) or 4
    buf = torch.zeros(nwg * nw * 64, dtype=torch.int64, device=fl.dev)
    Tp = ops.round_up(T, 8)
    ao = torch.randn(B, T, 512, device=fl.dev).to(fl.tdt)
    x = torch.randn(B, T, fl.C, device=fl.dev)
    if split:
        qk = torch.empty(B, T, 2048, dtype=torch.bfloat16, device=fl.dev)
        vt = torch.zeros(B, 2, 512, Tp, dtype=torch.bfloat16, device=fl.dev)
        ldq, vt_bs = 2048, 2 * 512 * Tp
    else:
        qk, vt = fl._new(B, T, 1024), torch.zeros(B, 512, Tp, dtype=fl.tdt, device=fl.dev)
        ldq, vt_bs = 1024, 512 * Tp
    for i in range(6):                                   # warm L2 / instruction cache; the last launch's stamps are read
        w, wn = blocks[i], blocks[i + 1]
        nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp,
                           vt_bs=vt_bs)
        if i == 5:
            assert lib.mmx_lab_tail_stamps(C.c_void_p(buf.data_ptr())) == 0
        ops.est_tail(ao, x, w, B=B, T=T, dtype=dt, bm=bm, nxt=nxt, waves=waves % 100, pf=pf, occ2=100 <= waves < 200, narrow=waves >= 200)
    torch.cuda.synchronize()
    assert lib.mmx_lab_tail_stamps(C.c_void_p(0)) == 0
    s = buf.cpu().reshape(nwg, nw, 64).double()
    names = {0: "entry", 1: "operand loads issued + tile copy", 2: "barrier", 3: "Wo MFMA", 4: "Wo epilogue", 5: "LN3 + A1 + barrier"}
    for ch in range(2):
        for h in range(4):
            names[6 + ch * 12 + 2 * h] = f"ch{ch} FF1 pass {h} MFMA"
            names[7 + ch * 12 + 2 * h] = f"ch{ch} FF1 pass {h} GELU epilogue"
        names[14 + ch * 12] = f"ch{ch} chunk barrier"
        names[15 + ch * 12] = f"ch{ch} FF2 MFMA"
        names[16 + ch * 12] = f"ch{ch} barrier"
    if split and bm == 64:                               # the 64-row split tile: four 256-wide chunks, five stamps each from 6 on
        for k in range(6, 31):
            names.pop(k, None)
        for ch in range(4):
            sb = 6 + ch * 5
            names[sb], names[sb + 1], names[sb + 2], names[sb + 3], names[sb + 4] = (f"ch{ch} FF1 MFMA", f"ch{ch} GELU epilogue", f"ch{ch} chunk barrier",
                                                                                   f"ch{ch} FF2 MFMA", f"ch{ch} barrier")
    names[31] = "closing epilogue (x store)"
    names[32] = "LN1 + A1 + barrier"
    for q in range(12):
        names[33 + 2 * q] = f"QKV pass {q} MFMA"
        names[34 + 2 * q] = f"QKV pass {q} epilogue"
    names[63] = "end"
    used = [i for i in range(64) if (s[:, :, i] > 0).all()]
    print(f"stamps: rows {2 * n * T}, {bm} rows x {nw} waves, pf {pf}, {nwg} workgroups; median over workgroups and waves, us (2.1 GHz ticks)")
    tot = 0.0
    for a, b in zip(used[:-1], used[1:]):
        d = ((s[:, :, b] - s[:, :, a]) / 2100.0).flatten()   # s_memtime: shader clock, ~2.1 GHz (tools/decode_lab.py)
        tot += d.median().item()
        print(f"   {names.get(b, b):38s} {d.median().item():7.2f}   (min {d.min().item():6.2f}, max {d.max().item():6.2f})   cumulative {tot:7.2f}")


if "--stamps" in sys.argv:
    if split and "--64" in sys.argv:
        stamps(1, 500, 64, 208, 2)
        stamps(5, 1000, 64, 208, 2)
        stamps(8, 896, 64, 208, 2)
    elif split:
        stamps(5, 1000, 32, 8, 0)
    else:
        stamps(5, 1000, 64, 4, 2)
        stamps(5, 1000, 64, 208, 4)
        stamps(5, 1000, 32, 8, 2)
else:
    sweep()
