"""Launches ONLY the dominant kernel (LM decode gate/up projection + SwiGLU, bf16) for PMC collection:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -- python3 tools/pmc_skinny.py
FETCH_SIZE on gfx950 counts 64 B per 128 B request for wide coalesced streams: double it (MI355X_MICROARCH.md §HBM)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, I, L = 896, 4864, 24
ws = [ops.pack_skinny((torch.randn(2 * I, H, device="cuda") / 30).bfloat16(), dtype=1, interleave_half=I) for _ in range(L)]
pk = B >= 4                                      # same activation layout as LlmEngine at this batch size
x = torch.randn(B, H, device="cuda").bfloat16()
if pk:
    x = ops.pack_act(x, 1)
act = torch.empty(ops.packed_rows(B), I, device="cuda", dtype=torch.bfloat16)
for it in range(4 * L):
    ops.skinny_gemm(x, ws[it % L], B=B, K=H, N=I, dtype=1, rs=True, epi=1, out_act=act, x_packed=pk, out_packed=pk)
torch.cuda.synchronize()
print("done")
