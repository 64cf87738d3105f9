"""Eager (no hipGraph) LM decode steps at batch 32 from context ~300, for rocprofv3 --pmc passes (counter collection does not
see kernels replayed from a hipGraph on this stack).    python tools/prof_decode.py [bf16|x] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import shapes, synth
from mmx.llm import LlmEngine
dt = {"bf16": 1, "x": 3, "f32": 0}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = 32
eng = LlmEngine(synth.synth_state_dict(shapes.llm_manifest(), 0), dtype=dt, max_batch=B, max_ctx=640, use_graphs=False)
g = torch.Generator().manual_seed(2)
z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
xs = [eng.build_lm_input(torch.randint(0, 151936, (1, 298), generator=g).cuda(), z, z) for _ in range(B)]
eng.start(xs, [steps + 4] * B, [steps + 4] * B, seed=0)
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
