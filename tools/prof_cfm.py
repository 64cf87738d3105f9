"""One batched CFM solve (n utterances x T frames) for kernel-trace profiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import shapes, synth
from mmx.flow import FlowEngine
n, T = int(sys.argv[1]), int(sys.argv[2])
dt = {"bf16": 1, "f32": 0, "x": 2}[sys.argv[3]] if len(sys.argv) > 3 else 1
eng = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=dt, use_graphs=False)
mus = [torch.randn(T, 80, device="cuda") for _ in range(n)]
conds = [torch.zeros(T, 80, device="cuda") for _ in range(n)]
spks = [torch.randn(80, device="cuda") for _ in range(n)]
eng.n_timesteps = 2
for _ in range(2):
    eng.cfm_batch(mus, spks, conds)
torch.cuda.synchronize()
