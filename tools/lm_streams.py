"""Does running two 16-sequence decode graphs concurrently on two streams beat one 32-sequence graph?  The decode step is
a chain of ~120 latency-bound kernels that leave most of the chip idle; two independent chains overlap their latencies
and pay for it with a second pass over the weights (HBM has the room)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch  # noqa: E402
from mmx import shapes, synth  # noqa: E402
from mmx.llm import LlmEngine  # noqa: E402

sd = synth.synth_state_dict(shapes.llm_manifest(), 0)
z = torch.zeros(1, 0, dtype=torch.long).cuda()
g = torch.Generator().manual_seed(0)


def begin(e):
    xs = [e.build_lm_input(torch.randint(0, 151936, (1, 48), generator=g).cuda(), z, z) for _ in range(e.B)]
    e.start(xs, [500] * e.B, [500] * e.B, seed=1)
    for _ in range(3):
        e.step()
    torch.cuda.synchronize()
    return e


def timed(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


e32 = begin(LlmEngine(sd, dtype=1, max_batch=32, max_ctx=640))
print("32 sequences, one stream          : %.3f ms/step" % timed(e32.step), flush=True)
a = LlmEngine(None, dtype=1, max_batch=16, max_ctx=640, share_from=e32)
b = LlmEngine(None, dtype=1, max_batch=16, max_ctx=640, share_from=e32)
for s_ in range(32):
    e32.release(s_)
begin(a), begin(b)
print("16 sequences, one stream          : %.3f ms/step" % timed(a.step), flush=True)
s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)


def pair():
    with torch.cuda.stream(s1):
        a.step()
    with torch.cuda.stream(s2):
        b.step()


print("2 x 16 sequences, two streams     : %.3f ms/step-pair" % timed(pair), flush=True)
