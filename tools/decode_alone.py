"""The decode loop of the config-4 rank share ALONE: the same tts_batch control flow (batched prefill, 32-slot decode,
compaction to the 16-slot engine, harvest / poll schedule, worker threads) with the flow + DAC stage replaced by
zero waveforms — a measurement tool, not a product switch.

    python tools/decode_alone.py [--dtype bf16|f32|x] [--per-gpu 32] [--steps 5] [--v1] [--cfg gu=2,1:down=2,8]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
from mmx import shapes, synth  # noqa: E402
from mmx.pipeline import TtsEngine  # noqa: E402


class DecodeOnly(TtsEngine):
    def _flow_dac_group(self, grp, toks, embs, wavs, frame_quantum, flow=None, prompts=None):
        for b in grp:
            wavs[b] = torch.zeros(1, 1, 2 * toks[b].numel() * self.hop, device=self.dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--per-gpu", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--v1", action="store_true", help="split build: the round-2 projection kernel instead of csrc/decode.hip")
    ap.add_argument("--cfg", default="", help="split build: LlmEngine.v2_cfg overrides, e.g. gu=1,1:down=2,8")
    ap.add_argument("--prefetch", type=int, default=0, help="workgroups of the per-layer weight prefetch on a side stream (LlmEngine.prefetch)")
    ap.add_argument("--lm-planes", default=None, choices=["f16x2", "bf16x3"])
    a = ap.parse_args()
    dt = {"bf16": 1, "f32": 0, "x": 2}[a.dtype]
    from mmx.llm import LlmEngine
    LlmEngine.use_v2 = not a.v1
    LlmEngine.prefetch = a.prefetch
    if a.lm_planes:
        LlmEngine.lm_planes = a.lm_planes
    if a.cfg:
        LlmEngine.v2_cfg = dict(LlmEngine.v2_cfg, **{kv.split("=")[0]: tuple(int(v) for v in kv.split("=")[1].split(",")) for kv in a.cfg.split(":")})
    eng = DecodeOnly(synth.synth_state_dict(shapes.llm_manifest(), 0), synth.synth_state_dict(shapes.flow_manifest(num_mid_blocks=1), 0),
                     synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0), dtype=dt, max_batch=a.per_gpu, max_ctx=640)
    lens = torch.randint(50, 501, (a.per_gpu,), generator=torch.Generator().manual_seed(3)).tolist()
    g = torch.Generator().manual_seed(2)
    texts = [torch.randint(0, 151936, (1, 48), generator=g).cuda() for _ in lens]
    emb = torch.randn(1, 192, generator=torch.Generator().manual_seed(1)).cuda()
    fn = lambda: eng.tts_batch(texts, [emb] * len(texts), seed=0, exact_steps=lens)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"decode loop alone, {a.per_gpu} utterances (max {max(lens)} steps), {a.dtype}: {ms:.1f} ms per step "
          f"({ms / max(lens):.3f} ms per decode step)", flush=True)
    # host cost of issuing one decode step (one hipGraph replay) vs the GPU time of the step: how far the host thread
    # of a rank is from being the bottleneck (8 ranks share one host)
    z = torch.zeros(1, 0, dtype=torch.long, device="cuda")
    xs = [eng.llm.build_lm_input(t, z, z) for t in texts]
    n = 200
    eng.llm.start(xs, [n + 8] * len(xs), [n + 8] * len(xs), seed=0)
    eng.llm.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.llm.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue time {(t1 - t0) / n * 1e6:.1f} us per decode step; GPU {(t2 - t0) / n * 1e6:.1f} us per decode step "
          f"(batch {len(xs)})", flush=True)


if __name__ == "__main__":
    main()
