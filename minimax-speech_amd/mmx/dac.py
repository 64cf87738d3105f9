"""DAC-VAE engines on libmmx_hip kernels: decoder (latents [B, D, T] -> 24 kHz waveform [B, 1, T*hop]) and encoder
(waveform [B, 1, T] -> (z, mu, logs) [B, D, T/hop]; SURVEY.md §8f row 3, the prompt-audio side of the path).

Reference op sequence: dac-vae/model.py:485-488 (DACVAE.decode), :326-379 (Decoder), :237-323 (DecoderBlock),
:107-143 (ResidualUnit), :509-514 (every Conv1d is followed by LeakyReLU(0.1)), layers.py:18-24 (Snake).

Mapping: every Conv1d / ConvTranspose1d is ONE windowed-GEMM launch (csrc/gemm.hip) on time-major
activations, with bias, LeakyReLU, the residual add and the NEXT layer's Snake fused into its epilogue;
the fp32 residual stream x and the compute-dtype activation snake(x) are the only tensors that touch HBM.
37 convs -> 37 launches + 1 transpose + 1 tail kernel.  Weight norm (w = g*v/||v||) is folded at load.
"""
import math
from typing import Dict, List

import torch

from . import ops
from ._lib import BF16, F32, TORCH_DT, X2, X2W


class DacDecoderEngine:
    FUSED_RU_MAX_C = 192          # ResidualUnits of at most this many channels run as ONE kernel (mmx_dac_ru: 48 / 96 / 192)

    def __init__(self, sd: Dict[str, torch.Tensor], rates: List[int], dtype=BF16, device="cuda", use_tanh=True,
                 with_pre=True, fuse_ru=True, wplanes=False):
        """wplanes (split build only): the checkpoint's folded fp32 weights are carried as two bf16 planes (MMX_X2W) instead of
        being rounded to bf16 - for checkpoints whose weights are not bf16-representable (any trained weight norm)."""
        self.dtype, self.tdt, self.dev = dtype, TORCH_DT[dtype], torch.device(device)
        self.wplanes = dtype == X2 and ops.resolve_wplanes(wplanes, (ops.fold_weight_norm(sd[k], sd[k[:-1] + "v"]) for k in sd
                                                                     if k.endswith(".weight_g") and not k.startswith("encoder.")))
        # the fused ResidualUnit kernel exists for the bf16 and split builds; the fp32 build keeps two GEMM launches per unit
        self.fuse_ru = bool(fuse_ru) and dtype in (BF16, X2)
        dtype = X2W if self.wplanes else dtype                         # `dtype` below: the code the weights are packed for
        self.rates = list(rates)
        self.hop = int(math.prod(rates))
        self.use_tanh = use_tanh
        f = lambda k: sd[k].detach().to(self.dev, torch.float32)
        wn = lambda p: ops.fold_weight_norm(f(p + ".weight_g"), f(p + ".weight_v"))
        bias = lambda p: f(p + ".bias").contiguous() if (p + ".bias") in sd else None
        alpha = lambda k: f(k).reshape(-1).contiguous()
        p = "decoder.model"
        self.D = sd[p + ".0.0.weight_v"].shape[1]
        self.with_pre = with_pre
        if with_pre:
            self.w_pre = ops.pack_conv1d(wn("de_conv_pre.0"), dtype)
            self.b_pre = bias("de_conv_pre.0")
        self.C0 = sd[p + ".0.0.weight_v"].shape[0]
        self.w0 = ops.pack_conv1d(wn(p + ".0.0"), dtype)
        self.b0 = bias(p + ".0.0")
        self.blocks = []
        n = len(rates)
        for i, s in enumerate(rates):
            q = f"{p}.{1 + i}.block"
            wt = wn(q + ".1")                                   # [Cin, Cout, 2s]
            blk = dict(stride=s, cin=wt.shape[0], cout=wt.shape[1], alpha_in=alpha(q + ".0.alpha"),
                       wt=ops.pack_convtranspose1d(wt, s, dtype), bt=bias(q + ".1"), rus=[])
            blk["fused"] = self.fuse_ru and blk["cout"] in (48, 96, 192) and blk["cout"] <= self.FUSED_RU_MAX_C
            for j, d in enumerate((1, 3, 9)):
                r = f"{q}.{2 + j}.block"
                w7, w1 = wn(r + ".1.0"), wn(r + ".3.0")
                ru = dict(dil=d, a0=alpha(r + ".0.alpha"), b7=bias(r + ".1.0"), a2=alpha(r + ".2.alpha"), b1=bias(r + ".3.0"))
                if blk["fused"]:
                    ru["w7_p"], ru["w1_p"] = ops.pack_dac_ru(w7, w1, dtype)
                else:
                    ru["w7"], ru["w1"] = ops.pack_conv1d(w7, dtype), ops.pack_conv1d(w1, dtype)
                blk["rus"].append(ru)
            self.blocks.append(blk)
        self.alpha_final = alpha(f"{p}.{n + 1}.alpha")
        wf = wn(f"{p}.{n + 2}.0")                                # [1, C, 7]
        assert wf.shape[0] == 1, "d_out != 1 is outside the hot path (configx2.yml: d_out 1)"
        self.k_final = wf.shape[2]
        self.w_final = wf[0].t().contiguous()                    # [k][C] fp32
        self.b_final = bias(f"{p}.{n + 2}.0")
        self.ctx_left, self.ctx_right = self.receptive_field(self.rates, self.k_final)

    @staticmethod
    def receptive_field(rates, k_final=7, k0=7, dils=(1, 3, 9)):
        """(left, right) latent frames an output sample can depend on beyond its own frame (dac-vae/model.py:326-379):
        final conv k7, per DecoderBlock three ResidualUnits (k7, dilation 1 / 3 / 9) behind a ConvTranspose1d
        (kernel 2s, stride s, padding ceil(s/2): output t reads inputs (t + p - 2s + 1)/s .. (t + p)/s), first conv k7.
        configx2.yml rates (5,4,4,3,2): 16 left, 15 right.  Streaming decodes a window with this much context."""
        L = R = (k_final - 1) // 2
        for s in reversed(list(rates)):
            for d in reversed(dils):
                L += 3 * d
                R += 3 * d
            p = math.ceil(s / 2)
            L, R = -((-(L + 2 * s - 1 - p)) // s), -((-(R + p)) // s)
        return L + (k0 - 1) // 2, R + (k0 - 1) // 2

    @torch.no_grad()
    def decode(self, z: torch.Tensor, skip_pre=False) -> torch.Tensor:
        """z [B, D, T] fp32 cuda -> [B, 1, T*hop] fp32 (reference layout). skip_pre: z is already de_conv_pre's output."""
        assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 3 and z.shape[1] == self.D
        B, D, T = z.shape
        z = z.contiguous()
        zt = torch.empty(B, T, D, dtype=self.tdt, device=self.dev)
        ops.copy2d(z, F32, D * T, 1, T, zt, self.dtype, T * D, D, 1, rows=T, cols=D, batch=B)
        return self.decode_time_major(zt, B, T, skip_pre)

    @torch.no_grad()
    def decode_time_major(self, zt: torch.Tensor, B: int, T: int, skip_pre=False, lens=None) -> torch.Tensor:
        """zt [B, T, D] in the compute dtype (what the flow engine hands over).
        lens (list of ints, optional): a ZERO-PADDED batch - member b has lens[b] <= T valid frames.  Every layer then treats the
        rows beyond a member's length as the zero padding a decode of that member alone would see: the windowed GEMMs multiply
        them by a row mask in their epilogue, a ConvTranspose1d's output (whose GEMM rows straddle the boundary) is masked by
        one row-mask launch, the fused ResidualUnits take the lengths themselves (mmx_dac_ru).  Member b of the result is
        wav[b, :, :lens[b] * hop], equal to decode_time_major(zt[b:b+1, :lens[b]]) bit for bit (the same arithmetic per row)."""
        dt, tdt, dev = self.dtype, self.tdt, self.dev
        new = lambda t, c, d=None: torch.empty(B, t, c, dtype=(d or tdt), device=dev)
        if lens is not None:
            assert len(lens) == B and max(lens) <= T
            ln = torch.tensor(lens, dtype=torch.int32, device=dev)
            T0 = T                                            # (T is rebound stage by stage below)
            mask_at = lambda rate: (torch.arange(T0 * rate, device=dev)[None, :] < (ln * rate)[:, None]).float().contiguous()
        rm, rate = (mask_at(1) if lens is not None else None), 1
        if skip_pre:
            h = zt
        else:
            h = new(T, self.D)
            ops.conv1d(zt, self.w_pre, T=T, Cin=self.D, k=1, dtype=dt, batch=B, bias=self.b_pre, act="lrelu", rowmask=rm, out_act=h)
        a = new(T, self.C0)
        ops.conv1d(h, self.w0, T=T, Cin=self.D, k=7, pad_left=3, dtype=dt, batch=B, bias=self.b0, act="lrelu", rowmask=rm,
                   alpha=self.blocks[0]["alpha_in"], out_act=a)
        for bi, blk in enumerate(self.blocks):
            s, cin, cout = blk["stride"], blk["cin"], blk["cout"]
            T2 = T * s
            rate *= s
            x = new(T2, cout, torch.float32)
            if blk["fused"]:
                # one kernel per ResidualUnit: the fp32 residual stream is all that passes between them; the last unit also
                # writes the next layer's input activation (a padded batch: the units zero the rows beyond a member's length on
                # the way in and out themselves)
                ops.convtranspose1d(a, blk["wt"], T=T, Cin=cin, Cout=cout, stride=s, dtype=dt, batch=B, bias=blk["bt"], out_f32=x)
                T = T2
                nxt = self.blocks[bi + 1]["alpha_in"] if bi + 1 < len(self.blocks) else self.alpha_final
                ln_s = (ln * rate).to(torch.int32) if lens is not None else None
                for j, ru in enumerate(blk["rus"]):
                    last = j == 2
                    x2 = new(T, cout, torch.float32)
                    a = new(T, cout) if last else None
                    ops.dac_ru(x, x2, ru, B=B, T=T, C_=cout, dil=ru["dil"], dtype=dt, act_out=a, alpha_next=(nxt if last else None), lens=ln_s)
                    x = x2
                continue
            a2 = new(T2, cout)
            ops.convtranspose1d(a, blk["wt"], T=T, Cin=cin, Cout=cout, stride=s, dtype=dt, batch=B, bias=blk["bt"],
                                alpha=blk["rus"][0]["a0"], out_f32=x, out_act=a2)
            T, a = T2, a2
            rm = mask_at(rate) if lens is not None else None
            if rm is not None:                             # the transposed conv's GEMM rows straddle a member's end: mask its outputs
                ops.mask_rows(x, rm, rows=B * T, C_=cout, dtype=F32)
                ops.mask_rows(a, rm, rows=B * T, C_=cout, dtype=dt)
            for j, ru in enumerate(blk["rus"]):
                d = ru["dil"]
                hmid = new(T, cout)
                ops.conv1d(a, ru["w7"], T=T, Cin=cout, k=7, dil=d, pad_left=3 * d, dtype=dt, batch=B, bias=ru["b7"],
                           act="lrelu", alpha=ru["a2"], rowmask=rm, out_act=hmid)
                last = j == 2
                if not last:
                    nxt = blk["rus"][j + 1]["a0"]
                elif bi + 1 < len(self.blocks):
                    nxt = self.blocks[bi + 1]["alpha_in"]
                else:
                    nxt = self.alpha_final
                x2 = None if last else new(T, cout, torch.float32)
                a3 = new(T, cout)
                ops.conv1d(hmid, ru["w1"], T=T, Cin=cout, k=1, dtype=dt, batch=B, bias=ru["b1"], act="lrelu",
                           residual=x, alpha=nxt, rowmask=rm, out_f32=x2, out_act=a3)
                x, a = x2, a3
        wav = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
        ops.conv_cout1_tanh(a, self.w_final, self.b_final, wav, T=T, C_=self.blocks[-1]["cout"], k=self.k_final,
                            batch=B, dtype=dt, use_tanh=self.use_tanh)
        return wav


class DacEncoderEngine:
    """DACVAE.encode (dac-vae/model.py:469-483): Encoder (:195-234) -> leaky_relu(0.01) -> en_conv_post -> (m | logs) -> z.

    Same mapping as the decoder: one windowed-GEMM launch per Conv1d, the strided down-sampling convs
    (EncoderBlock, :182-188: k=2s, stride s, pad ceil(s/2)) through MmxGemmParams.row_stride; the d_in=1 head is its
    own kernel (K=7 is no GEMM).  31 GEMM launches + head + VAE-sampling kernel + 3 layout copies."""

    def __init__(self, sd: Dict[str, torch.Tensor], rates: List[int], dtype=BF16, device="cuda"):
        self.dtype, self.tdt, self.dev = dtype, TORCH_DT[dtype], torch.device(device)
        self.rates = list(rates)
        self.hop = int(math.prod(rates))
        f = lambda k: sd[k].detach().to(self.dev, torch.float32)
        wn = lambda p: ops.fold_weight_norm(f(p + ".weight_g"), f(p + ".weight_v"))
        bias = lambda p: f(p + ".bias").contiguous()
        alpha = lambda k: f(k).reshape(-1).contiguous()
        p = "encoder.block"
        w0 = wn(p + ".0.0")                                     # [C0, 1, 7]
        assert w0.shape[1] == 1, "d_in != 1 is outside the hot path (configx2.yml: d_in 1)"
        self.C0, self.k0 = w0.shape[0], w0.shape[2]
        self.w0 = w0.reshape(self.C0, self.k0).contiguous()
        self.b0 = bias(p + ".0.0")
        self.blocks = []
        c = self.C0
        for i, s in enumerate(rates):
            q = f"{p}.{1 + i}.block"
            blk = dict(stride=s, c=c, rus=[], alpha_out=alpha(q + ".3.alpha"), wd=ops.pack_conv1d(wn(q + ".4.0"), dtype),
                       bd=bias(q + ".4.0"))
            for j, d in enumerate((1, 3, 9)):
                r = f"{q}.{j}.block"
                blk["rus"].append(dict(dil=d, a0=alpha(r + ".0.alpha"), w7=ops.pack_conv1d(wn(r + ".1.0"), dtype),
                                       b7=bias(r + ".1.0"), a2=alpha(r + ".2.alpha"),
                                       w1=ops.pack_conv1d(wn(r + ".3.0"), dtype), b1=bias(r + ".3.0")))
            self.blocks.append(blk)
            c *= 2
        n = len(rates)
        self.C_last = c
        self.alpha_final = alpha(f"{p}.{n + 1}.alpha")
        self.w_final = ops.pack_conv1d(wn(f"{p}.{n + 2}.0"), dtype)   # [D, 3*C]
        self.b_final = bias(f"{p}.{n + 2}.0")
        self.D = self.w_final.shape[0]
        self.w_post = ops.pack_conv1d(wn("en_conv_post.0"), dtype)    # [2D, D]
        self.b_post = bias("en_conv_post.0")

    @staticmethod
    def _down(T, s):
        """Output length of Conv1d(k=2s, stride=s, padding=ceil(s/2)) (model.py:182-188)."""
        return (T + 2 * math.ceil(s / 2) - 2 * s) // s + 1

    def frames(self, T: int) -> int:
        """Latent frames for T input samples."""
        for s in self.rates:
            T = self._down(T, s)
            if T < 1:
                return 0
        return T

    @torch.no_grad()
    def encode(self, audio: torch.Tensor, noise: torch.Tensor = None, generator: torch.Generator = None):
        """audio [B, 1, T] fp32 cuda -> (z, mu, logs), each [B, D, T'] fp32; T' = T/hop for a hop multiple
        (DACVAE.preprocess), else each stride-s conv floors as torch's Conv1d does (extract_dac_latents.py:20-36
        encodes unpadded audio).  `noise` [B, D, T'] stands for the reference's torch.randn_like(m); drawn on the
        device when absent."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 3 and audio.shape[1] == 1
        B, _, T = audio.shape
        if self.frames(T) < 1:
            raise ValueError(f"audio of {T} samples is shorter than one latent frame")
        dt, tdt, dev = self.dtype, self.tdt, self.dev
        new = lambda t, c, d=None: torch.empty(B, t, c, dtype=(d or tdt), device=dev)
        audio = audio.contiguous()
        x = new(T, self.C0, torch.float32)
        a = new(T, self.C0)
        ops.conv_cin1(audio, self.w0, self.b0, T=T, C_=self.C0, k=self.k0, batch=B, dtype=dt,
                      alpha=self.blocks[0]["rus"][0]["a0"], out_f32=x, out_act=a)
        for bi, blk in enumerate(self.blocks):
            s, c = blk["stride"], blk["c"]
            for j, ru in enumerate(blk["rus"]):
                d = ru["dil"]
                hmid = new(T, c)
                ops.conv1d(a, ru["w7"], T=T, Cin=c, k=7, dil=d, pad_left=3 * d, dtype=dt, batch=B, bias=ru["b7"],
                           act="lrelu", alpha=ru["a2"], out_act=hmid)
                last = j == 2
                nxt = blk["alpha_out"] if last else blk["rus"][j + 1]["a0"]
                x2 = None if last else new(T, c, torch.float32)
                a3 = new(T, c)
                ops.conv1d(hmid, ru["w1"], T=T, Cin=c, k=1, dtype=dt, batch=B, bias=ru["b1"], act="lrelu",
                           residual=x, alpha=nxt, out_f32=x2, out_act=a3)
                x, a = x2, a3
            T2 = self._down(T, s)
            nxt = self.blocks[bi + 1]["rus"][0]["a0"] if bi + 1 < len(self.blocks) else self.alpha_final
            last_blk = bi + 1 == len(self.blocks)
            x = None if last_blk else new(T2, 2 * c, torch.float32)
            a2 = new(T2, 2 * c)
            ops.conv1d(a, blk["wd"], T=T, Cin=c, k=2 * s, pad_left=math.ceil(s / 2), stride=s, T_out=T2, dtype=dt,
                       batch=B, bias=blk["bd"], act="lrelu", alpha=nxt, out_f32=x, out_act=a2)
            T, a = T2, a2
        # conv k3 + LeakyReLU(0.1), then F.leaky_relu(0.01) (model.py:475): one LeakyReLU of slope 0.1*0.01
        h = new(T, self.D)
        ops.conv1d(a, self.w_final, T=T, Cin=self.C_last, k=3, pad_left=1, dtype=dt, batch=B, bias=self.b_final,
                   act="lrelu", slope=0.1 * 0.01, out_act=h)
        ml = new(T, 2 * self.D, torch.float32)
        ops.conv1d(h, self.w_post, T=T, Cin=self.D, k=1, dtype=dt, batch=B, bias=self.b_post, act="lrelu", out_f32=ml)
        D = self.D
        if noise is None:
            noise_t = torch.randn(B, T, D, dtype=torch.float32, device=dev, generator=generator)
        else:
            assert noise.shape == (B, D, T)
            noise_t = torch.empty(B, T, D, dtype=torch.float32, device=dev)
            noise = noise.to(dev, torch.float32).contiguous()
            ops.copy2d(noise, F32, D * T, 1, T, noise_t, F32, T * D, D, 1, rows=T, cols=D, batch=B)
        zt, mt, lt = (torch.empty(B, T, D, dtype=torch.float32, device=dev) for _ in range(3))
        ops.vae_sample(ml, noise_t, zt, mt, lt, rows=B * T, D=D)
        outs = []
        for src in (zt, mt, lt):                               # -> the reference's channels-first [B, D, T]
            o = torch.empty(B, D, T, dtype=torch.float32, device=dev)
            ops.copy2d(src, F32, T * D, D, 1, o, F32, D * T, 1, T, rows=T, cols=D, batch=B)
            outs.append(o)
        return tuple(outs)
