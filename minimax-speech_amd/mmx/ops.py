"""Tensor-level wrappers over the C ABI (one function per entry point) + weight packing helpers.

Layout convention: activations are time-major [B, T, C] contiguous tensors; "act" tensors have the
compute dtype (bf16 or fp32), residual streams are fp32.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from ._lib import ACT, BF16, F32, check, i64, load, stream, _p


def round_up(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- weight packing (load time)
class Planed(torch.Tensor):
    """A packed weight held as bf16 PLANES hi + [mid +] lo = w of an fp32 checkpoint's values (include/mmx_hip.h MMX_X2W / MMX_X3W).
    The type is the marker: gemm() / skinny2() switch to the weight-plane dtype code when they are handed one; views and row
    slices keep the type (torch's default __torch_function__)."""


def weight_planes(w: torch.Tensor, n: int):
    """fp32 -> n bf16 terms, round to nearest each: hi = bf16(w), mid = bf16(w - hi), ... (the remainders are exact in fp32)."""
    out, r = [], w.float()
    for _ in range(n):
        t = r.to(torch.bfloat16)
        out.append(t)
        r = r - t.float()
    return out


def resolve_wplanes(wplanes, weights) -> bool:
    """wplanes of an engine constructor: True / False, or "auto" = True exactly when one of `weights` (the matrix-shaped weights
    as the kernels will see them, weight norm folded) is not bf16-representable, i.e. when rounding at load would change it by
    more than 2^-16 relative - the precision the split build's products keep anyway (a weight-norm fold g * v / ||v|| with
    g = ||v|| leaves ~1e-6 on a bf16-representable v: the norm of 10 000 elements is not summed in the same order twice)."""
    if wplanes != "auto":
        return bool(wplanes)
    for w in weights:
        w = w.detach().float()
        if bool(((w - w.to(torch.bfloat16).float()).abs() > 2.0 ** -16 * w.abs()).any()):
            return True
    return False


def pack_linear(w: torch.Tensor, dtype) -> torch.Tensor:
    """[N, K] -> [N, Kpad] (K contiguous, zero padded to a multiple of 32) in the compute dtype.
    dtype X2W / X3W: w fp32 -> Planed [N, planes * Kpad], the planes side by side in every row."""
    N, K = w.shape
    Kp = round_up(K, 32)
    planes = L.WPLANES.get(dtype, 1)
    out = torch.zeros(N, planes * Kp, dtype=L.WEIGHT_DT[dtype], device=w.device)
    if planes == 1:
        out[:, :K] = w
        return out
    for i, t in enumerate(weight_planes(w, planes)):
        out[:, i * Kp:i * Kp + K] = t
    return out.as_subclass(Planed)


def conv1d_matrix(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, k] -> the GEMM's weight matrix [Cout, k*Cin] (tap-major, then channel), same dtype."""
    Cout, Cin, k = w.shape
    return w.permute(0, 2, 1).reshape(Cout, k * Cin)


def pack_conv1d(w: torch.Tensor, dtype) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, k] -> GEMM weight [Cout, k*Cin] (tap-major, then channel)."""
    return pack_linear(conv1d_matrix(w), dtype)


def pack_convtranspose1d(w: torch.Tensor, stride: int, dtype) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, 2s] -> [s*Cout, 2*Cin]: row (k0, co); column tap*Cin + ci with
    tap 0 multiplying x[q-1] (kernel index k0 + s) and tap 1 multiplying x[q] (kernel index k0)."""
    Cin, Cout, k = w.shape
    s = stride
    assert k == 2 * s
    wk = w.permute(2, 1, 0)                      # [k, Cout, Cin]
    tap0 = wk[s:2 * s]                           # [s, Cout, Cin]
    tap1 = wk[0:s]
    g = torch.cat([tap0, tap1], dim=2)           # [s, Cout, 2*Cin]
    return pack_linear(g.reshape(s * Cout, 2 * Cin), dtype)


def fold_weight_norm(g, v):
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * (v / n)


# ----------------------------------------------------------------------------- GEMM family
def gemm(A, W, M, N, *, dtype, lda=None, cin=None, ntaps=1, dil=1, row_off=0, row_lo=0, row_hi=None, batch=1,
         a_bstride=0, w_bstride=0, bias=None, bias_mod=None, bias_per_row=False, act="none", act2="none", slope=0.1,
         residual=None, ldr=0, r_bstride=0, rowmask=None, rm_bstride=0, alpha=None, alpha_mod=None,
         out_f32=None, ldo_f=0, of_bstride=0, out_act=None, ldo_a=0, oa_bstride=0, out_off=0, out_len=None, row_stride=1,
         tile=0):
    """Launches mmx_gemm_win. A/W/out_* may be tensors or raw device addresses (ints)."""
    ldw = W.shape[-1] if hasattr(W, "shape") else None
    assert ldw is not None
    if isinstance(W, Planed):                            # weight planes of an fp32 checkpoint
        dtype = {L.X2: L.X2W, L.X3: L.X3W, L.X2W: L.X2W, L.X3W: L.X3W}[dtype]
    elif dtype in L.WPLANES:
        raise L.MmxError("a weight-plane dtype needs a Planed weight (ops.pack_linear with X2W / X3W)")
    p = L.gemm_params(A=A, W=W, bias=bias, residual=residual, rowmask=rowmask, alpha=alpha, out_f32=out_f32,
                      out_act=out_act, lda=lda, ldw=ldw, ldr=ldr, ldo_f=ldo_f, ldo_a=ldo_a,
                      a_bstride=a_bstride, w_bstride=w_bstride, r_bstride=r_bstride, rm_bstride=rm_bstride,
                      of_bstride=of_bstride, oa_bstride=oa_bstride, row_off=row_off, row_lo=row_lo,
                      row_hi=(row_hi if row_hi is not None else (1 << 40)), out_off=out_off,
                      out_len=(out_len if out_len is not None else (1 << 62)), M=M, N=N, batch=batch,
                      ntaps=ntaps, cin=cin, dil=dil, bias_mod=(bias_mod or N), alpha_mod=(alpha_mod or N),
                      bias_per_row=int(bias_per_row), act=ACT[act], act2=ACT[act2], slope=slope, row_stride=row_stride)
    L.gemm_win(p, dtype, tile)


def linear(x, Wp, K, *, dtype, bias=None, act="none", act2="none", residual=None, rowmask=None, out_f32=None,
           out_act=None):
    """x: act tensor [..., K] (contiguous rows); Wp packed [N, Kpad]. Outputs are [..., N] tensors."""
    N = Wp.shape[0]
    M = x.numel() // x.shape[-1]
    gemm(x, Wp, M, N, dtype=dtype, lda=x.shape[-1], cin=K, bias=bias, act=act, act2=act2,
         residual=residual, ldr=N, rowmask=rowmask, out_f32=out_f32, ldo_f=N, out_act=out_act, ldo_a=N)


def conv1d(x, Wp, *, T, Cin, k, dtype, dil=1, pad_left=0, batch=1, bias=None, act="none", slope=0.1, residual=None,
           rowmask=None, alpha=None, out_f32=None, out_act=None, T_out=None, act2="none", stride=1):
    """Conv1d over time-major x [B, T, Cin] -> [B, T_out, Cout]; zero padding is virtual; stride via row_stride."""
    Cout = Wp.shape[0]
    T_out = T if T_out is None else T_out
    gemm(x, Wp, T_out, Cout, dtype=dtype, lda=Cin, cin=Cin, ntaps=k, dil=dil, row_off=-pad_left, row_lo=0, row_hi=T, row_stride=stride,
         batch=batch, a_bstride=T * Cin, bias=bias, act=act, act2=act2, slope=slope,
         residual=residual, ldr=Cout, r_bstride=T_out * Cout, rowmask=rowmask, rm_bstride=T_out,
         alpha=alpha, out_f32=out_f32, ldo_f=Cout, of_bstride=T_out * Cout,
         out_act=out_act, ldo_a=Cout, oa_bstride=T_out * Cout)


def convtranspose1d(x, Wp, *, T, Cin, Cout, stride, dtype, batch=1, bias=None, alpha=None, out_f32=None, out_act=None):
    """ConvTranspose1d(kernel 2s, stride s, padding ceil(s/2), output_padding s%2): [B,T,Cin] -> [B,T*s,Cout]."""
    s = stride
    pad = math.ceil(s / 2)
    gemm(x, Wp, T + 1, s * Cout, dtype=dtype, lda=Cin, cin=Cin, ntaps=2, dil=1, row_off=-1, row_lo=0, row_hi=T,
         batch=batch, a_bstride=T * Cin, bias=bias, bias_mod=Cout, alpha=alpha, alpha_mod=Cout,
         out_f32=out_f32, ldo_f=s * Cout, of_bstride=T * s * Cout, out_act=out_act, ldo_a=s * Cout,
         oa_bstride=T * s * Cout, out_off=-pad * Cout, out_len=T * s * Cout)


# ----------------------------------------------------------------------------- row-wise / elementwise
def rownorm(x, gamma, beta, eps, *, rows, C_, batch=1, x_bstride=None, rms=False, act="none", rowmask=None,
            addvec=None, av_bstride=None, out_f32=None, out_act=None, dtype=F32, ldx=None, o_bstride=None, rm_bstride=None):
    ldx = C_ if ldx is None else ldx
    xb = rows * ldx if x_bstride is None else x_bstride
    ob = rows * C_ if o_bstride is None else o_bstride
    check(load().mmx_rownorm(_p(x), i64(ldx), i64(xb), rows, C_, batch, _p(gamma), _p(beta), C.c_float(eps), int(rms),
                             ACT[act], _p(rowmask), i64(rows if rm_bstride is None else rm_bstride), _p(addvec),
                             i64(C_ if av_bstride is None else av_bstride), _p(out_f32), i64(C_), i64(ob), _p(out_act), i64(C_),
                             i64(ob), dtype, stream()), "mmx_rownorm")


def groupnorm(x, gamma, beta, out, *, B, T, C_, groups, dtype, eps=1e-5, act="none", rowmask=None):
    check(load().mmx_groupnorm(_p(x), B, T, C_, groups, _p(gamma), _p(beta), C.c_float(eps), ACT[act], _p(rowmask), _p(out),
                               dtype, stream()), "mmx_groupnorm")


def act_rows(x, *, rows, C_, act="none", rowmask=None, out_f32=None, out_act=None, dtype=F32):
    check(load().mmx_act_rows(_p(x), i64(rows), C_, ACT[act], _p(rowmask), _p(out_f32), _p(out_act), dtype, stream()),
          "mmx_act_rows")


def mask_rows(x, rowmask, *, rows, C_, dtype):
    """x[row][:] = 0 where rowmask[row] == 0, in place (mmx_mask_rows)."""
    check(load().mmx_mask_rows(_p(x), i64(rows), C_, _p(rowmask), dtype, stream()), "mmx_mask_rows")


def gather_rows(ids, table, *, scale=1.0, rowmask=None, out_f32=None, out_act=None, dtype=F32):
    n, C_ = ids.numel(), table.shape[1]
    check(load().mmx_gather_rows(_p(ids), n, _p(table), C_, C.c_float(scale), _p(rowmask), _p(out_f32), i64(C_),
                                 _p(out_act), i64(C_), dtype, stream()), "mmx_gather_rows")


def copy2d(src, src_dt, ibs, irs, ics, dst, dst_dt, obs, ors, ocs, rows, cols, batch=1, rep=1):
    check(load().mmx_copy2d(_p(src), src_dt, i64(ibs), i64(irs), i64(ics), rep, _p(dst), dst_dt, i64(obs), i64(ors),
                            i64(ocs), rows, cols, batch, stream()), "mmx_copy2d")


def est_pack(x, mu, spks, cond, h, *, B, T, dtype, x_bstride=None, x_mod=None):
    check(load().mmx_est_pack(_p(x), i64(T * 80 if x_bstride is None else x_bstride), (B if x_mod is None else x_mod), _p(mu), _p(spks), _p(cond), B, T, 80, _p(h), i64(h.shape[-1]), dtype, stream()),
          "mmx_est_pack")


def sinusoidal_emb(t, out, *, dim, dtype, scale=1000.0):
    check(load().mmx_sinusoidal_emb(_p(t), t.numel(), dim, C.c_float(scale), _p(out), dtype, stream()),
          "mmx_sinusoidal_emb")


def cfg_euler(x, d_cond, d_uncond, cfg, dt, n):
    check(load().mmx_cfg_euler(_p(x), _p(d_cond), _p(d_uncond), C.c_float(cfg), C.c_float(dt), i64(n), stream()),
          "mmx_cfg_euler")


def conv_cin1(x, w, bias, *, T, C_, k, batch, dtype, slope=0.1, alpha=None, out_f32=None, out_act=None):
    check(load().mmx_conv_cin1(_p(x), i64(T), T, C_, k, _p(w), _p(bias), C.c_float(slope), _p(alpha), _p(out_f32), _p(out_act),
                               batch, dtype, stream()), "mmx_conv_cin1")


def vae_sample(ml, noise, z, m, logs, *, rows, D):
    check(load().mmx_vae_sample(_p(ml), _p(noise), i64(rows), D, _p(z), _p(m), _p(logs), stream()), "mmx_vae_sample")


def prefetch4(tensors, sink, workgroups=128):
    """mmx_prefetch4: leaves the bytes of up to four tensors in L2 / the Infinity Cache (a side stream of the LM decode step)."""
    t = list(tensors) + [None] * (4 - len(tensors))
    nb = lambda x: 0 if x is None else (x.numel() * x.element_size()) // 16 * 16
    check(load().mmx_prefetch4(_p(t[0]), i64(nb(t[0])), _p(t[1]), i64(nb(t[1])), _p(t[2]), i64(nb(t[2])), _p(t[3]), i64(nb(t[3])),
                               _p(sink), workgroups, stream()), "mmx_prefetch4")


def resample_linear(x, T2):
    """x fp32 [..., T] -> [..., T2]: F.interpolate(mode="linear", align_corners=False) along the last axis (mmx_resample_linear)."""
    x = x.contiguous()
    T = x.shape[-1]
    out = torch.empty(*x.shape[:-1], T2, dtype=torch.float32, device=x.device)
    check(load().mmx_resample_linear(_p(x), i64(x.numel() // T), T, T2, _p(out), stream()), "mmx_resample_linear")
    return out


def conv_cout1_tanh(act, w, bias, out, *, T, C_, k, batch, dtype, slope=0.1, use_tanh=True):
    check(load().mmx_conv_cout1_tanh(_p(act), i64(T * C_), T, C_, k, _p(w), _p(bias), C.c_float(slope), int(use_tanh),
                                     _p(out), i64(T), batch, dtype, stream()), "mmx_conv_cout1_tanh")


# ----------------------------------------------------------------------------- fused estimator row-tile kernels
def est_next(wqkv=None, n1g=None, n1b=None, q_out=None, ldq=0, q_bs=0, vt_out=None, ldvt=0, vt_bs=0):
    return L.fill_struct(L.EstNext(), wqkv=wqkv, n1g=n1g, n1b=n1b, q_out=q_out, vt_out=vt_out, q_bs=q_bs, vt_bs=vt_bs,
                         ldq=ldq, ldvt=ldvt)


def est_tail(ao, x, w, *, B, T, dtype, bm, rowmask=None, act_out=None, act_ld=0, nxt=None, eps=1e-5, pf=0, t_begin=0,
             Tcap=None, waves=0, occ2=False, narrow=False):
    """w: dict with packed wo_p / w1_p / w2_p and bo / b1 / b2 / n3g / n3b (mmx/flow.py).  Tcap: frames every buffer is
    allocated for (batch stride; default T); t_begin: first frame to process (streaming hop)."""
    Tc = T if Tcap is None else Tcap
    p = L.fill_struct(L.EstTailParams(), ao=ao, x=x, wo=w["wo_p"], w1=w["w1_p"], w2=w["w2_p"], bo=w["bo"], b1=w["b1"],
                      b2=w["b2"], n3g=w["n3g"], n3b=w["n3b"], rowmask=rowmask, act_out=act_out, ao_bs=Tc * 512,
                      x_bs=Tc * 256, rm_bs=Tc, act_bs=Tc * act_ld, ldao=512, act_ld=act_ld, B=B, T=T, t_begin=t_begin, eps=eps)
    if nxt is not None:
        p.next = nxt
    if isinstance(w["wo_p"], Planed):                   # [hi pack | lo pack] weights (pack_skinny with X2W), the next block's too
        dtype = L.X2W
    check(load().mmx_est_tail(C.byref(p), C.c_int(dtype), C.c_int(bm), C.c_int(pf + 16 * waves + (256 if occ2 else 0) + (512 if narrow else 0)), stream()),
          "mmx_est_tail")


def est_resnet(a_in, lda, cin, x, r, tv, tv_bs, *, B, T, dtype, bm, rowmask=None, nxt=None, eps=1e-5, pf=0, t_begin=0,
               Tcap=None, waves=0):
    Tc = T if Tcap is None else Tcap
    p = L.fill_struct(L.EstResnetParams(), a_in=a_in, x=x, w1=r["w1_p"], w2=r["w2_p"], wr=r["wr_p"], b1=r["b1"], g1=r["g1"],
                      be1=r["be1"], b2=r["b2"], g2=r["g2"], be2=r["be2"], br=r["br"], tv=tv, rowmask=rowmask,
                      a_bs=Tc * lda, x_bs=Tc * 256, tv_bs=tv_bs, rm_bs=Tc, lda=lda, cin=cin, B=B, T=T, t_begin=t_begin, eps=eps)
    if nxt is not None:
        p.next = nxt
    if isinstance(r["w1_p"], Planed):
        dtype = L.X2W
    check(load().mmx_est_resnet(C.byref(p), C.c_int(dtype), C.c_int(bm), C.c_int(pf + 16 * waves), stream()), "mmx_est_resnet")


# ----------------------------------------------------------------------------- fused DAC ResidualUnit
def pack_dac_ru(w7: torch.Tensor, w1: torch.Tensor, dtype):
    """ResidualUnit conv weights [C, C, 7] and [C, C, 1] (weight norm folded) -> fragment-ordered packs for mmx_dac_ru:
    [C][7 * CP] tap-major with each tap's input channels zero-padded to CP = 32 * ceil(C / 32), and [C][CP]."""
    C_ = w7.shape[0]
    CP = (C_ + 31) // 32 * 32
    wd = torch.float32 if dtype in L.WPLANES else L.WEIGHT_DT[dtype]     # weight planes: pack_skinny splits the fp32 values
    a = torch.zeros(C_, 7, CP, dtype=torch.float32, device=w7.device)
    a[:, :, :C_] = w7.permute(0, 2, 1)
    b = torch.zeros(C_, CP, dtype=torch.float32, device=w7.device)
    b[:, :C_] = w1[:, :, 0]
    return (pack_skinny(a.reshape(C_, 7 * CP).to(wd).contiguous(), dtype=dtype), pack_skinny(b.to(wd).contiguous(), dtype=dtype))


def dac_ru(x, x_out, ru, *, B, T, C_, dil, dtype, act_out=None, alpha_next=None, lens=None, bm=0, slope=0.1, x_bs=None):
    """ru: dict with w7_p / w1_p (pack_dac_ru), b7 / b1 / a0 / a2 (mmx/dac.py).  include/mmx_hip.h: mmx_dac_ru."""
    p = L.fill_struct(L.DacRuParams(), x=x, x_out=x_out, act_out=act_out, w7=ru["w7_p"], w1=ru["w1_p"], b7=ru["b7"], b1=ru["b1"],
                      a0=ru["a0"], a2=ru["a2"], alpha_next=alpha_next, lens=lens, x_bs=(T * C_ if x_bs is None else x_bs),
                      B=B, T=T, C=C_, dil=dil, slope=slope)
    if isinstance(ru["w7_p"], Planed):
        dtype = L.X2W
    check(load().mmx_dac_ru(C.byref(p), C.c_int(dtype), C.c_int(bm), stream()), "mmx_dac_ru")


# ----------------------------------------------------------------------------- attention
def attn_dense(q, k, v, out, *, B, H, Tq, Tk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale, dtype,
               keymask=None, chunk=0, pos=None, ldp=0, pos_u=None, pos_v=None, head_stride=0, q_begin=0, km_bs=None):
    check(load().mmx_attn_dense(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(v), i64(ldv), i64(v_bs),
                                _p(out), i64(ldo), i64(o_bs), B, H, 64, Tq, Tk, C.c_float(scale), _p(keymask),
                                i64(Tk if km_bs is None else km_bs), chunk, _p(pos), i64(ldp), _p(pos_u), _p(pos_v), head_stride,
                                q_begin, dtype, stream()),
          "mmx_attn_dense")


def attn_relpos_bf16(q, k, vt, pos, pos_u, pos_v, out, *, B, H, T, ldq, ldk, ldvt, ldp, ldo, q_bs, k_bs, vt_bs, o_bs, scale, chunk=0,
                     klen=None):
    if klen is not None:
        assert klen.dtype == torch.int32 and klen.numel() >= B
    check(load().mmx_attn_relpos_bf16(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(vt), i64(ldvt), i64(vt_bs),
                                      _p(pos), i64(ldp), _p(pos_u), _p(pos_v), _p(out), i64(ldo), i64(o_bs), B, H, T,
                                      C.c_float(scale), chunk, _p(klen), stream()), "mmx_attn_relpos_bf16")


def attn_relpos_x(q, k, v, pos, pos_u, pos_v, out, *, B, H, T, ldq, ldk, ldv, ldp, ldo, q_bs, k_bs, v_bs, o_bs, scale, chunk=0, klen=None):
    """The split build's conformer rel-pos attention on the MFMA (include/mmx_hip.h mmx_attn_relpos_x): fp32 operands."""
    if klen is not None:
        assert klen.dtype == torch.int32 and klen.numel() >= B
    check(load().mmx_attn_relpos_x(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(v), i64(ldv), i64(v_bs), _p(pos), i64(ldp),
                                   _p(pos_u), _p(pos_v), _p(out), i64(ldo), i64(o_bs), B, H, T, C.c_float(scale), chunk, _p(klen), stream()),
          "mmx_attn_relpos_x")


def attn_flash_bf16(q, k, vt, out, *, B, H, T, ldq, ldk, ldvt, ldo, q_bs, k_bs, vt_bs, o_bs, scale, keymask=None,
                    chunk=0, q_begin=0, km_bs=None, fp8=False, klen=None):
    """klen: int32 [B] valid keys per batch row (prefix masks of a padded batch) - see include/mmx_hip.h."""
    if klen is not None:
        assert klen.dtype == torch.int32 and klen.numel() >= B
    if fp8:
        check(load().mmx_attn_flash_fp8(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(vt), i64(ldvt),
                                        i64(vt_bs), _p(out), i64(ldo), i64(o_bs), B, H, T, C.c_float(scale),
                                        _p(keymask), i64(T if km_bs is None else km_bs), chunk, q_begin, _p(klen), stream()), "mmx_attn_flash_fp8")
        return
    check(load().mmx_attn_flash_bf16(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(vt), i64(ldvt),
                                     i64(vt_bs), _p(out), i64(ldo), i64(o_bs), B, H, T, C.c_float(scale),
                                     _p(keymask), i64(T if km_bs is None else km_bs), chunk, q_begin, _p(klen), stream()), "mmx_attn_flash_bf16")


def attn_flash_x(q, k, v, out, *, B, H, T, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale, keymask=None, chunk=0, q_begin=0,
                 km_bs=None, klen=None):
    """The split build's flash attention: fp32 q / k / v (V row-major) and fp32 out (include/mmx_hip.h)."""
    if klen is not None:
        assert klen.dtype == torch.int32 and klen.numel() >= B
    check(load().mmx_attn_flash_x(_p(q), i64(ldq), i64(q_bs), _p(k), i64(ldk), i64(k_bs), _p(v), i64(ldv), i64(v_bs),
                                  _p(out), i64(ldo), i64(o_bs), B, H, T, C.c_float(scale), _p(keymask),
                                  i64(T if km_bs is None else km_bs), chunk, q_begin, _p(klen), stream()), "mmx_attn_flash_x")


def attn_flash_xs(qk, vt, out, *, B, H, T, ldqk, ldvt, ldo, qk_bs, vt_bs, o_bs, scale, keymask=None, chunk=0, q_begin=0, km_bs=None,
                  klen=None, form=0):
    """The split build's flash attention on pre-split operands (include/mmx_hip.h): qk bf16 [B, T, >=2048], vt bf16 [B, 2, 512, ldvt].
    form: 0 = workgroup shape chosen per launch, 1 = the co-residency-friendly shape (launches beside the decode loop)."""
    if klen is not None:
        assert klen.dtype == torch.int32 and klen.numel() >= B
    check(load().mmx_attn_flash_xs(_p(qk), i64(ldqk), i64(qk_bs), _p(vt), i64(ldvt), i64(vt_bs), _p(out), i64(ldo), i64(o_bs), B, H, T,
                                   C.c_float(scale), _p(keymask), i64(T if km_bs is None else km_bs), chunk, q_begin, _p(klen), int(form),
                                   stream()), "mmx_attn_flash_xs")


# ----------------------------------------------------------------------------- LM decode
def pack_skinny(w, *, dtype, kscale=None, interleave_half=0):
    """w: [N, K] tensor in the weight dtype -> MFMA-fragment-ordered copy (see csrc/llm.hip).
    dtype X3W (mmx_skinny2 only): w fp32 -> the packs of its three bf16 planes one after the other (Planed)."""
    N, K = w.shape
    if dtype in L.WPLANES:
        assert w.dtype == torch.float32 and kscale is None
        base = {L.X2W: L.X2, L.X3W: L.X3}[dtype]
        return torch.cat([pack_skinny(t.contiguous(), dtype=base, interleave_half=interleave_half)
                          for t in weight_planes(w, L.WPLANES[dtype])]).as_subclass(Planed)
    assert w.dtype == L.WEIGHT_DT[dtype]
    KB = 32 if L.WEIGHT_DT[dtype] == torch.bfloat16 else 16
    tiles = (N + 15) // 16
    wp = torch.empty(tiles * (K // KB) * 64 * (KB // 4), dtype=L.WEIGHT_DT[dtype], device=w.device)
    check(load().mmx_pack_skinny(_p(w), i64(w.stride(0)), N, K, _p(kscale), interleave_half, _p(wp), dtype, stream()),
          "mmx_pack_skinny")
    return wp


X_PACKED, OUT_PACKED = 1, 2


def packed_rows(B):
    """Rows a packed activation buffer holds for batch B (whole 16-row MFMA tiles)."""
    return (B + 15) // 16 * 16


def pack_act(x, dtype):
    """Host-side statement of the packed activation layout of include/mmx_hip.h (tests / tools): x [B, K] -> flat
    xp[m][kb][lane][E]."""
    E = 8 if dtype == BF16 else 4
    B, K = x.shape
    R = packed_rows(B)
    xp = torch.zeros(R, K, dtype=x.dtype, device=x.device)
    xp[:B] = x
    return xp.reshape(R // 16, 16, K // (4 * E), 4, E).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def unpack_act(xp, B, K, dtype):
    E = 8 if dtype == BF16 else 4
    R = packed_rows(B)
    return xp.reshape(R // 16, K // (4 * E), 4, 16, E).permute(0, 3, 1, 2, 4).reshape(R, K)[:B]


def skinny_gemm(x, wp, *, B, K, N, dtype, bias=None, rs=False, eps=1e-6, epi=0, out_f32=None, out_act=None,
                ldx=None, ldo_f=None, ldo_a=None, x_packed=False, out_packed=False, kgamma=None):
    """kgamma (split builds only): the RMSNorm gain [K], applied to x in the kernel instead of being folded into wp."""
    xdt = L.dt_of(x)
    flags = (X_PACKED if x_packed else 0) | (OUT_PACKED if out_packed else 0)
    check(load().mmx_skinny_gemm(_p(x), xdt, i64(ldx if ldx is not None else K), B, K, N, _p(wp), _p(bias), int(rs),
                                 C.c_float(eps), epi, _p(out_f32), i64(ldo_f if ldo_f is not None else N),
                                 _p(out_act), i64(ldo_a if ldo_a is not None else N), dtype, flags, _p(kgamma), stream()),
          "mmx_skinny_gemm")


SSQ_SLOTS = 64


def plane_elems(B, K):
    """bf16 elements of ONE plane of a split-plane activation [B, K] (include/mmx_hip.h)."""
    return packed_rows(B) * K


def split_planes(x, f16=False):
    """Host-side statement of the split-plane format (tests): x fp32 [B, K] -> bf16 [3, plane_elems] (hi + mid + lo), or with
    f16 the two fp16 planes hi + lo of MMX_H2, returned as their bit patterns in a bf16-typed tensor [2, plane_elems]."""
    B, K = x.shape
    out, r = [], x.float()
    for _ in range(2 if f16 else 3):
        t = r.to(torch.float16 if f16 else torch.bfloat16)
        out.append(pack_act(t.view(torch.bfloat16) if f16 else t, BF16))
        r = r - t.float()
    return torch.stack(out)


def merge_planes(xs, B, K, f16=False):
    """bf16 [3, plane_elems] -> fp32 [B, K] (hi + mid + lo); f16: the two fp16 planes (bit patterns in a bf16-typed tensor)."""
    if f16:
        return sum(unpack_act(xs[s], B, K, BF16).contiguous().view(torch.float16).float() for s in range(2))
    return sum(unpack_act(xs[s], B, K, BF16).float() for s in range(3))


H2_WSCALE = 256.0            # csrc/decode.hip: the fp16 forms store weights * 2^8


def pack_skinny_h2(w, *, planes=1, interleave_half=0):
    """[N, K] fp32 weights -> the MMX_H2 (planes = 1: for bf16-representable checkpoints, exact) or MMX_H2W (planes = 2: hi + lo of
    an fp32 checkpoint, Planed) pack of mmx_skinny2: fp16 values of w * 2^8 in the MFMA-fragment order (the pack is a
    permutation of 16-bit words, so the bf16 packer carries the fp16 bit patterns)."""
    ws = w.float() * H2_WSCALE
    assert float(ws.abs().max()) < 65504.0, "weight out of the fp16 range of MMX_H2"
    out, r = [], ws
    for _ in range(planes):
        t = r.to(torch.float16)
        out.append(pack_skinny(t.view(torch.bfloat16).contiguous(), dtype=L.X3, interleave_half=interleave_half))
        r = r - t.float()
    return out[0] if planes == 1 else torch.cat(out).as_subclass(Planed)


def decode_prep(x, xs, ssq, *, B, K, gamma=None, h=None, dtype=L.X3):
    check(load().mmx_decode_prep(_p(x), i64(x.shape[-1]), B, K, _p(gamma), _p(h), i64(h.shape[-1] if h is not None else 0), _p(xs),
                                 _p(ssq), dtype, stream()), "mmx_decode_prep")


def skinny2(xs, wp, *, B, K, N, dtype, bias=None, ssq_in=None, eps=1e-6, epi=0, out=None, ldo=None, xs_out=None, gamma_next=None,
            ssq_out=None, tiles_per_wg=1, ksplit=1, part=None, tickets=None):
    """The split build's decode-step projection on split-plane activations (include/mmx_hip.h mmx_skinny2)."""
    if isinstance(wp, Planed):
        dtype = L.H2W if dtype in (L.H2, L.H2W) else L.X3W
    check(load().mmx_skinny2(_p(xs), B, K, N, _p(wp), _p(bias), _p(ssq_in), C.c_float(eps), epi, _p(out),
                             i64(ldo if ldo is not None else N), _p(xs_out), _p(gamma_next), _p(ssq_out), tiles_per_wg, ksplit,
                             _p(part), i64(part.numel() if part is not None else 0), _p(tickets), dtype, stream()), "mmx_skinny2")


def rope_kv_store(qkv, inv_freq, pos, q_out, kc, vc, block_table, *, B, rows, Hq, Hkv, page, dtype):
    ld = (Hq + 2 * Hkv) * 64
    check(load().mmx_rope_kv_store(_p(qkv), i64(ld), i64(rows * ld), B, rows, Hq, Hkv, 64, _p(inv_freq), _p(pos),
                                   _p(q_out), i64(Hq * 64), i64(rows * Hq * 64), _p(kc), _p(vc), _p(block_table),
                                   block_table.shape[1], page, dtype, stream()), "mmx_rope_kv_store")


def paged_attn(q, pos, kc, vc, block_table, out, *, B, rows, Hq, Hkv, page, dtype):
    check(load().mmx_paged_attn(_p(q), i64(Hq * 64), i64(rows * Hq * 64), B, rows, Hq, Hkv, 64, C.c_float(0.125),
                                _p(pos), _p(kc), _p(vc), _p(block_table), block_table.shape[1], page, _p(out),
                                i64(Hq * 64), i64(rows * Hq * 64), dtype, stream()), "mmx_paged_attn")


def decode_attn(qkv, inv_freq, pos, kc, vc, block_table, out, *, B, Hq, Hkv, page, dtype, rope_tab=None, out_packed=False,
                per_head=False, out_split=False, one_head=False):
    """per_head: the per-query-head kernel even where the GQA-shared one applies (measurements, tests); one_head: that kernel with
    one head per workgroup instead of two (the round-3 form; identical results).
    out_split: `out` receives split planes (the split build's decode step, include/mmx_hip.h): True = three bf16 planes,
    "f16" = two fp16 planes (MMX_H2)."""
    check(load().mmx_decode_attn(_p(qkv), i64((Hq + 2 * Hkv) * 64), B, Hq, Hkv, 64, _p(inv_freq), _p(rope_tab), _p(pos), _p(kc),
                                 _p(vc), _p(block_table), block_table.shape[1], page, C.c_float(0.125), _p(out),
                                 i64(Hq * 64), dtype, int(bool(out_packed)) | (2 if per_head else 0) | (8 if out_split == "f16" else (4 if out_split else 0)) | (16 if one_head else 0), stream()), "mmx_decode_attn")


def swiglu(gu, out, *, rows, I, dtype):
    check(load().mmx_swiglu(_p(gu), i64(2 * I), rows, I, _p(out), i64(I), dtype, stream()), "mmx_swiglu")


def sample_step(logits, state, out_tokens, speech_emb, next_x, *, V, B, eos_id, seed, top_k=25, top_p=0.8,
                win_size=10, tau_r=0.1, sampled=None, forced=None, logp_out=None):
    check(load().mmx_sample_step(_p(logits), i64(logits.shape[-1]), V, B, eos_id, top_k, C.c_float(top_p), win_size,
                                 C.c_float(tau_r), C.c_uint64(seed), _p(state), _p(out_tokens), out_tokens.shape[1],
                                 _p(sampled), _p(forced), _p(speech_emb), speech_emb.shape[1], _p(next_x),
                                 i64(next_x.shape[-1]), _p(logp_out), stream()), "mmx_sample_step")
