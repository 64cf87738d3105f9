"""mmx_dac_ru launch time by stage (channels, rows per 10 s of audio), dilation and tile height, against the two windowed-
GEMM launches it replaces.    python tools/dac_lab.py [bf16|x] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
sys.path.insert(0, ROOT)
from mmx import ops, shapes, synth  # noqa: E402
from mmx.dac import DacDecoderEngine  # noqa: E402
from bench import _event_time_graph  # noqa: E402

dt = {"bf16": 1, "x": 2}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = synth.synth_state_dict(shapes.dac_decoder_manifest(80), 0)
fused, plain = DacDecoderEngine(sd, [5, 4, 4, 3, 2], dtype=dt), DacDecoderEngine(sd, [5, 4, 4, 3, 2], dtype=dt, fuse_ru=False)
T = 500
for bi, blk in enumerate(fused.blocks):
    T *= blk["stride"]
    C_ = blk["cout"]
    if not blk["fused"]:
        continue
    x = torch.randn(B, T, C_, device="cuda")
    x2 = torch.empty_like(x)
    a = torch.empty(B, T, C_, dtype=fused.tdt, device="cuda")
    h = torch.empty_like(a)
    a3 = torch.empty_like(a)
    for j, ru in enumerate(blk["rus"]):
        pr = plain.blocks[bi]["rus"][j]
        d = ru["dil"]

        def two(i=0):
            ops.conv1d(a, pr["w7"], T=T, Cin=C_, k=7, dil=d, pad_left=3 * d, dtype=dt, batch=B, bias=pr["b7"], act="lrelu", alpha=pr["a2"], out_act=h)
            ops.conv1d(h, pr["w1"], T=T, Cin=C_, k=1, dtype=dt, batch=B, bias=pr["b1"], act="lrelu", residual=x, alpha=pr["a0"], out_f32=x2, out_act=a3)

        line = [f"two launches {2 * _event_time_graph(two, 10):6.1f}"]
        for bm in (256, 128, 64, 32):
            try:
                us = _event_time_graph(lambda i=0: ops.dac_ru(x, x2, ru, B=B, T=T, C_=C_, dil=d, dtype=dt, bm=bm), 20)
                line.append(f"bm {bm}: {us:6.1f}")
            except Exception:  # noqa: BLE001
                pass
        gb = 2 * B * T * C_ * 4 / 1e9
        print(f"C {C_:3d} rows {B * T:7d} dil {d}: " + " | ".join(line) + f"   (x in + out {gb * 1e3:.0f} MB: {gb / 4e3 * 1e6:.0f} us at 4 TB/s)", flush=True)
