"""Does running two B=16 decode graphs concurrently on two streams beat one B=32 graph? (latency-bound kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import shapes, synth
from mmx.llm import LlmEngine

sd = synth.synth_state_dict(shapes.llm_manifest(), 0)
z = torch.zeros(1, 0, dtype=torch.long).cuda()
g = torch.Generator().manual_seed(0)


def mk(B, share=None):
    e = LlmEngine(sd if share is None else None, dtype=1, max_batch=B, max_ctx=640, share_from=share)
    if share is not None:   # own KV pages for an independent batch
        e.kc = torch.zeros_like(share.kc[:, : B * e.max_pages + 1]); e.vc = torch.zeros_like(e.kc)
        e.block_table = torch.arange(B * e.max_pages, dtype=torch.int32, device="cuda").reshape(B, e.max_pages).contiguous()
        e.trash_page = B * e.max_pages
    xs = [e.build_lm_input(torch.randint(0, 151936, (1, 48), generator=g).cuda(), z, z) for _ in range(B)]
    e.start(xs, [500] * B, [500] * B, seed=1)
    for _ in range(3):
        e.step()
    torch.cuda.synchronize()
    return e


def timed(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


e32 = mk(32)
print("B=32 one stream      : %.3f ms/step" % timed(e32.step))
a = mk(16, share=e32); b = mk(16, share=e32)
print("B=16 one stream      : %.3f ms/step" % timed(a.step))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def pair():
    with torch.cuda.stream(s1): a.step()
    with torch.cuda.stream(s2): b.step()
print("2 x B=16, two streams: %.3f ms/step-pair" % timed(pair))
c = [mk(8, share=e32) for _ in range(4)]; ss = [torch.cuda.Stream() for _ in range(4)]
def quad():
    for e, s in zip(c, ss):
        with torch.cuda.stream(s): e.step()
print("4 x B=8, four streams: %.3f ms/step-quad" % timed(quad))
