"""Times the fused estimator kernels (est_tail, est_resnet) per launch for each tile height / weight-ring depth at a
batched shape and at the single-utterance shape (hipGraph of launches rotating over the 48 mid blocks' weights)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
sys.path.insert(0, ROOT)
from mmx import ops, shapes, synth  # noqa: E402
from mmx.flow import FlowEngine  # noqa: E402
from bench import _event_time_graph  # noqa: E402


def main():
    fl = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=1, parts=("estimator",))
    blocks = [w for st in fl.mid for w in st["blocks"]]
    res = [st["res"] for st in fl.mid]
    for B, T in ((16, 896), (2, 500), (4, 500)):
        Tp = ops.round_up(T, 8)
        ao = torch.randn(B, T, 512, device="cuda").bfloat16()
        a_in = torch.randn(B, T, 256, device="cuda").bfloat16()
        x = torch.randn(B, T, 256, device="cuda")
        tv = torch.randn(B, 14 * 256, device="cuda")
        qk, vt = torch.empty(B, T, 1024, device="cuda", dtype=torch.bfloat16), torch.zeros(B, 512, Tp, device="cuda", dtype=torch.bfloat16)
        M = B * T
        for bm, pf, wv in ((64, 2, 4), (64, 2, 8), (32, 4, 4), (32, 4, 8), (32, 2, 8), (16, 8, 4), (16, 4, 8)):
            if True:
                def tail(i=0):
                    w, wn = blocks[i % len(blocks)], blocks[(i + 1) % len(blocks)]
                    nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=1024, q_bs=T * 1024, vt_out=vt, ldvt=Tp, vt_bs=512 * Tp)
                    ops.est_tail(ao, x, w, B=B, T=T, dtype=1, bm=bm, nxt=nxt, pf=pf, waves=wv)

                def resn(i=0):
                    r, wn = res[i % len(res)], blocks[i % len(blocks)]
                    nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=1024, q_bs=T * 1024, vt_out=vt, ldvt=Tp, vt_bs=512 * Tp)
                    ops.est_resnet(a_in, 256, 256, x, r, tv, 14 * 256, B=B, T=T, dtype=1, bm=bm, nxt=nxt, pf=pf, waves=wv)

                ut, ur = _event_time_graph(tail, 96), _event_time_graph(resn, 96)
                ft = 2.0 * M * (512 * 256 + 2 * 256 * 1024 + 256 * 1536)
                print(f"B={B:2d} T={T:4d} bm={bm:2d} pf={pf} waves={wv}: tail {ut:7.2f} us ({ft / ut / 1e6:6.1f} TF)   resnet {ur:7.2f} us", flush=True)


if __name__ == "__main__":
    main()
