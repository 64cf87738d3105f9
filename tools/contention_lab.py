"""How much one kind of flow launch, replayed continuously on a normal-priority stream, slows the LM decode step that runs
on the high-priority stream beside it (batch 32, context ~300).    python tools/contention_lab.py [bf16|x]"""
import gc
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
from mmx import ops, shapes, synth  # noqa: E402
from mmx.flow import FlowEngine  # noqa: E402
from mmx.llm import LlmEngine  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
SPLIT = len(sys.argv) > 1 and sys.argv[1] == "x"
FDT = 2 if SPLIT else 1
llm = LlmEngine(synth.synth_state_dict(shapes.llm_manifest(), 0), dtype=(3 if SPLIT else 1), max_batch=B, max_ctx=768)
fl = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=FDT, use_graphs=False)
blocks = [w for st in fl.mid for w in st["blocks"]]
g = torch.Generator().manual_seed(2)
z = torch.zeros(1, 0, dtype=torch.long, device=dev)
xs = [llm.build_lm_input(torch.randint(0, 151936, (1, 298), generator=g).cuda(), z, z) for _ in range(B)]
hi, side = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)


def graph_of(fn, iters):
    fn(0)
    gr = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    gc.collect()
    with torch.cuda.graph(gr):
        for i in range(iters):
            fn(i)
    return gr


def tail_graph(n, T, bm, **cfg):
    Bf, Tp = 2 * n, ops.round_up(T, 8)
    ao = torch.randn(Bf, T, 512, device=dev).to(fl.tdt)
    x = torch.randn(Bf, T, fl.C, device=dev)
    if SPLIT:
        qk = torch.empty(Bf, T, 2048, dtype=torch.bfloat16, device=dev)
        vt = torch.zeros(Bf, 2, 512, Tp, dtype=torch.bfloat16, device=dev)
        ldq, vt_bs = 2048, 2 * 512 * Tp
    else:
        qk, vt = fl._new(Bf, T, 1024), torch.zeros(Bf, 512, Tp, dtype=fl.tdt, device=dev)
        ldq, vt_bs = 1024, 512 * Tp

    def one(i=0):
        w, wn = blocks[i % len(blocks)], blocks[(i + 1) % len(blocks)]
        nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp, vt_bs=vt_bs)
        ops.est_tail(ao, x, w, B=Bf, T=T, dtype=FDT, bm=bm, nxt=nxt, **cfg)
    return graph_of(one, 112)


def flash_graph(n, T, form=0):
    Bf, Tp = 2 * n, ops.round_up(T, 8)
    if SPLIT:
        qk = torch.randn(Bf, T, 2048, device=dev).bfloat16()
        vt = torch.randn(Bf, 2, 512, Tp, device=dev).bfloat16()
        ao = torch.empty(Bf, T, 512, device=dev)
        return graph_of(lambda i=0: ops.attn_flash_xs(qk, vt, ao, B=Bf, H=8, T=T, ldqk=2048, ldvt=Tp, ldo=512, qk_bs=T * 2048, vt_bs=2 * 512 * Tp,
                                                      o_bs=T * 512, scale=0.125, form=form), 112)
    qk = torch.randn(Bf, T, 1024, device=dev).to(fl.tdt)
    vt = torch.randn(Bf, 512, Tp, device=dev).to(fl.tdt)
    ao = torch.empty(Bf, T, 512, device=dev, dtype=fl.tdt)
    return graph_of(lambda i=0: ops.attn_flash_bf16(qk, qk[:, :, 512:], vt, ao, B=Bf, H=8, T=T, ldq=1024, ldk=1024, ldvt=Tp, ldo=512, q_bs=T * 1024,
                                                    k_bs=T * 1024, vt_bs=512 * Tp, o_bs=T * 512, scale=0.125), 112)


def resnet_graph(n, T, waves):
    res = [st["res"] for st in fl.mid]
    Bf, Tp = 2 * n, ops.round_up(T, 8)
    a_in = torch.randn(Bf, T, 256, device=dev); x = torch.randn(Bf, T, 256, device=dev); tv = torch.randn(Bf, 14 * 256, device=dev)
    qk = torch.empty(Bf, T, 2048, dtype=torch.bfloat16, device=dev); vt = torch.zeros(Bf, 2, 512, Tp, dtype=torch.bfloat16, device=dev)

    def one(i=0):
        r, wn = res[i % len(res)], blocks[i % len(blocks)]
        nxt = ops.est_next(wqkv=wn["wqkv_p"], n1g=wn["n1g"], n1b=wn["n1b"], q_out=qk, ldq=2048, q_bs=T * 2048, vt_out=vt, ldvt=Tp, vt_bs=2 * 512 * Tp)
        ops.est_resnet(a_in, 256, 256, x, r, tv, 14 * 256, B=Bf, T=T, dtype=2, bm=32, nxt=nxt, waves=waves)
    return graph_of(one, 96)


def measure(name, gr):
    llm.start(xs, [300] * B, [300] * B, seed=0)
    with torch.cuda.stream(hi):
        for _ in range(8):
            llm.step()
    torch.cuda.synchronize()
    stop = [False]

    def bg():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            while not stop[0]:
                gr.replay()
                side.synchronize()
    th = threading.Thread(target=bg, daemon=True) if gr is not None else None
    if th:
        th.start()
        time.sleep(0.05)
    with torch.cuda.stream(hi):
        hi.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            llm.step()
        hi.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
    stop[0] = True
    if th:
        th.join()
    torch.cuda.synchronize()
    print(f"{name:52s} decode step {us:7.1f} us", flush=True)


measure("alone", None)
if SPLIT:
    measure("beside est_tail 5x1000, 32 rows x 8 waves (320 wg)", tail_graph(5, 1000, 32))
    measure("beside est_tail 5x1000, 64 rows (160 wg)", tail_graph(5, 1000, 64))
    measure("beside est_tail 2x1000, 32 rows (128 wg)", tail_graph(2, 1000, 32))
    measure("beside est_tail 2x1000, 64 rows (64 wg)", tail_graph(2, 1000, 64))
    measure("beside flash_xs 5x1000 (launch-time form)", flash_graph(5, 1000))
    measure("beside flash_xs 5x1000, form 1 (128-query wgs)", flash_graph(5, 1000, 1))
    measure("beside flash_xs 2x1000", flash_graph(2, 1000))
    measure("beside est_resnet 5x1000, 32 rows x 4 waves", resnet_graph(5, 1000, 4))
    measure("beside est_resnet 5x1000, 32 rows x 8 waves", resnet_graph(5, 1000, 8))
    measure("alone again", None)
    sys.exit(0)
measure("beside est_tail 5x1000, 64 rows x 8 waves (160 wg)", tail_graph(5, 1000, 64))
measure("beside est_tail 5x1000, 64 rows x 4 waves (160 wg)", tail_graph(5, 1000, 64, waves=4, pf=2))
measure("beside est_tail 5x1000, 32 rows x 8 waves (320 wg)", tail_graph(5, 1000, 32))
measure("beside est_tail 2x1000, 64 rows (64 wg)", tail_graph(2, 1000, 64))
measure("beside flash 5x1000", flash_graph(5, 1000))
measure("alone again", None)
