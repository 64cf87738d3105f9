"""BASELINE config 2: the CFM decoder alone - 250 FSQ tokens -> 500 latent frames, 10 Euler steps with CFG, bf16, hipGraph
replay; also the whole flow.inference (token embedding + conformer encoder + CFM)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
from mmx import shapes, synth  # noqa: E402
from mmx.flow import FlowEngine  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--unfused", action="store_true", help="one launch per Linear / Conv1d / LayerNorm (the round-1 path)")
    a = ap.parse_args()
    eng = FlowEngine(synth.synth_state_dict(shapes.flow_manifest(), 0), dtype=(1 if a.dtype == "bf16" else 0), fused=(False if a.unfused else None))
    print(f"dtype {a.dtype}, {'unfused' if a.unfused else 'fused row-tile'} estimator", flush=True)
    g = torch.Generator().manual_seed(0)
    for n_tok in (250, 500):
        tok = torch.randint(0, 6561, (1, n_tok), generator=g).cuda()
        z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
        zf = torch.zeros(1, 0, 80, device="cuda")
        emb = torch.randn(1, 192, generator=g).cuda()
        mu, spks, cond, _ = eng.conditions(tok, z, zf, emb)
        t_cfm = timeit(lambda: eng.cfm(mu, spks, cond, False))
        t_all = timeit(lambda: eng.inference_time_major(tok, z, zf, emb))
        sec = n_tok / 25.0
        print(f"{n_tok} tokens ({sec:.0f} s of audio): CFM alone {t_cfm:.2f} ms ({sec / t_cfm * 1e3:.0f} audio-s/s), "
              f"flow.inference {t_all:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
