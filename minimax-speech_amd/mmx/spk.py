"""Learnable speaker encoder engine: reference mel crops -> L2-normalised 192-d embedding.

Reference: speech/cosyvoice/llm/llm.py:34-96 (LearnableSpeakerEncoder), transformer/arch_util.py:21-123 (GroupNorm32,
QKVAttentionLegacy, AttentionBlock).  One-shot cost per utterance: k1 convs are GEMM launches, GroupNorm its own kernel,
the head-interleaved legacy attention runs on the dense attention kernel with a head stride of 3*64 columns."""
from typing import Dict

import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT


class SpeakerEncoderEngine:
    def __init__(self, sd: Dict[str, torch.Tensor], dtype=BF16, device="cuda", prefix="speaker_encoder", heads=8, pack_dtype=None):
        self.dtype, self.tdt, self.dev, self.heads = dtype, TORCH_DT[dtype], torch.device(device), heads
        dtype = dtype if pack_dtype is None else pack_dtype     # the code the weights are packed for (X2W: weight planes)
        f = lambda k: sd[k].detach().to(self.dev, torch.float32).contiguous()
        self.w_init, self.b_init = ops.pack_conv1d(f(prefix + ".init.weight"), dtype), f(prefix + ".init.bias")
        self.C = sd[prefix + ".init.weight"].shape[0]
        self.mel_dim = sd[prefix + ".init.weight"].shape[1]
        self.blocks = []
        i = 0
        while f"{prefix}.attn.{i}.norm.weight" in sd:
            p = f"{prefix}.attn.{i}"
            self.blocks.append(dict(g=f(p + ".norm.weight"), b=f(p + ".norm.bias"),
                                    wqkv=ops.pack_conv1d(f(p + ".qkv.weight"), dtype), bqkv=f(p + ".qkv.bias"),
                                    wo=ops.pack_conv1d(f(p + ".proj_out.weight"), dtype), bo=f(p + ".proj_out.bias")))
            i += 1
        self.w_out, self.b_out = ops.pack_linear(f(prefix + ".output_proj.weight"), dtype), f(prefix + ".output_proj.bias")
        self.out_dim = sd[prefix + ".output_proj.weight"].shape[0]
        import math
        self.gamma_l2 = torch.full((self.out_dim,), 1.0 / math.sqrt(self.out_dim), device=self.dev)

    @torch.no_grad()
    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, 80, T] fp32 -> [B, 192] fp32, L2-normalised (LearnableSpeakerEncoder.forward, first-frame pooling)."""
        dt, C, H = self.dtype, self.C, self.heads
        B, M, T = mel.shape
        mel = mel.to(self.dev, torch.float32).contiguous()
        mt = torch.empty(B, T, M, dtype=self.tdt, device=self.dev)
        ops.copy2d(mel, F32, M * T, 1, T, mt, dt, T * M, M, 1, rows=T, cols=M, batch=B)
        x = torch.empty(B, T, C, device=self.dev)
        ops.conv1d(mt, self.w_init, T=T, Cin=M, k=1, dtype=dt, batch=B, bias=self.b_init, out_f32=x)
        hn = torch.empty(B, T, C, dtype=self.tdt, device=self.dev)
        qkv = torch.empty(B, T, 3 * C, dtype=self.tdt, device=self.dev)
        att = torch.empty(B, T, C, dtype=self.tdt, device=self.dev)
        for w in self.blocks:
            ops.groupnorm(x, w["g"], w["b"], hn, B=B, T=T, C_=C, groups=32, dtype=dt)
            ops.conv1d(hn, w["wqkv"], T=T, Cin=C, k=1, dtype=dt, batch=B, bias=w["bqkv"], out_act=qkv)
            # arch_util.py:62-66: heads are split first, then (q, k, v) inside a head -> head stride 3*64 columns
            ops.attn_dense(qkv, qkv[:, :, 64:], qkv[:, :, 128:], att, B=B, H=H, Tq=T, Tk=T, ldq=3 * C, ldk=3 * C, ldv=3 * C,
                           ldo=C, q_bs=T * 3 * C, k_bs=T * 3 * C, v_bs=T * 3 * C, o_bs=T * C, scale=0.125, dtype=dt,
                           head_stride=192)
            x2 = torch.empty(B, T, C, device=self.dev)
            ops.conv1d(att, w["wo"], T=T, Cin=C, k=1, dtype=dt, batch=B, bias=w["bo"], residual=x, out_f32=x2)
            x = x2
        # frame 0 of every batch item -> Linear -> L2 normalise
        first = torch.empty(B, C, dtype=self.tdt, device=self.dev)
        ops.copy2d(x, F32, T * C, C, 1, first, dt, C, C, 1, rows=1, cols=C, batch=B)
        y = torch.empty(B, self.out_dim, device=self.dev)
        ops.linear(first, self.w_out, C, dtype=dt, bias=self.b_out, out_f32=y)
        out = torch.empty(B, self.out_dim, device=self.dev)
        ops.rownorm(y, self.gamma_l2, None, 1e-30, rows=B, C_=self.out_dim, rms=True, out_f32=out, dtype=F32)
        return out

    @torch.no_grad()
    def reference_embedding(self, reference_mels: torch.Tensor) -> torch.Tensor:
        """flow.py:338-368 / llm.py:166-184: [B, N, 80, T] -> mean over the N references -> L2 normalise; or [B, 80, T]."""
        if reference_mels.dim() == 4:
            B, N, M, T = reference_mels.shape
            e = self.encode(reference_mels.reshape(B * N, M, T)).reshape(B, N, -1).mean(dim=1).contiguous()
        else:
            e = self.encode(reference_mels)
        out = torch.empty_like(e)
        ops.rownorm(e, self.gamma_l2, None, 1e-30, rows=e.shape[0], C_=self.out_dim, rms=True, out_f32=out, dtype=F32)
        return out
