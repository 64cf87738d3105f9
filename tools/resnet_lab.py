"""est_resnet launch time, split build: 4 waves vs 8 waves per workgroup (cin = 256), a few group shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd")); sys.path.insert(0, ROOT)
from mmx import ops, shapes, synth
from mmx.flow import FlowEngine
from bench import _event_time_graph
fl = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=2, use_graphs=False)
blocks = [w for st in fl.mid for w in st["blocks"]]
res = [st["res"] for st in fl.mid]
for n, T in [(1, 150), (1, 500), (2, 1000), (4, 1000), (5, 1000), (8, 896), (12, 1000)]:
    B = 2 * n
    Tp = ops.round_up(T, 8)
    a_in = torch.randn(B, T, 256, device="cuda"); x = torch.randn(B, T, 256, device="cuda"); tv = torch.randn(B, 14 * 256, device="cuda")
    qk = torch.empty(B, T, 2048, dtype=torch.bfloat16, device="cuda"); vt = torch.zeros(B, 2, 512, Tp, dtype=torch.bfloat16, device="cuda")
    line = []
    for bm, wv, pf in ((32, 4, 0), (32, 8, 0), (16, 4, 0), (16, 8, 0)):
        def resn(i=0):
            r, wn = res[i % len(res)], blocks[i % len(blocks)]
            nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=2048, q_bs=T * 2048, vt_out=vt, ldvt=Tp, vt_bs=2 * 512 * Tp)
            ops.est_resnet(a_in, 256, 256, x, r, tv, 14 * 256, B=B, T=T, dtype=2, bm=bm, nxt=nxt, pf=pf, waves=wv)
        line.append(f"{bm}x{wv}w: {_event_time_graph(resn, 96):6.1f}")
    print(f"rows {B * T:6d} | " + " | ".join(line), flush=True)
