"""Op-level parity tests of the HIP kernels (through the C ABI) against plain torch fp32 math.
Tolerances: fp32 build ~1e-4 relative (exact-fp32 MFMA, different summation order); bf16 build is
bounded by bf16 operand rounding (2^-8 relative per operand) with fp32 accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from mmx import _lib, ops
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    _lib.load()
    return _lib, ops


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


DT = [0, 1]   # F32, BF16
# 2 / 3 = the split builds (bf16 weights, fp32 activations as 2 / 3 bf16 terms per MFMA product): the reference below
# multiplies the fp32 activations with the bf16-rounded weights, so what is left is 2^-17 (X2) / fp32-level (X3) rounding
DTX = [0, 1, 2, 3]
TOL = {0: 2e-5, 1: 2e-2, 2: 4e-5, 3: 2e-5}


def cast(x, dt):
    return x.to(torch.bfloat16 if dt == 1 else torch.float32)


@pytest.mark.parametrize("dt", DTX)
@pytest.mark.parametrize("M,N,K", [(1, 16, 32), (37, 80, 96), (500, 256, 320), (130, 1536, 256), (1000, 48, 336), (64, 64, 4864)])
def test_gemm_plain(env, dt, M, N, K):
    L, ops = env
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda() / math.sqrt(K)
    b = torch.randn(N, generator=g).cuda()
    xa, wp = cast(x, dt), ops.pack_linear(w, dt)
    out = torch.zeros(M, N, device="cuda")
    outa = torch.zeros(M, N, device="cuda", dtype=xa.dtype)
    ops.linear(xa, wp, K, dtype=dt, bias=b, act="gelu", out_f32=out, out_act=outa)
    ref = F.gelu(F.linear(xa.float(), wp[:, :K].float(), b))
    assert rel_err(out, ref) < TOL[dt]
    assert rel_err(outa, ref) < (1e-2 if dt == 1 else TOL[dt])


@pytest.mark.parametrize("dt", DTX)
def test_gemm_epilogue_residual_mask_snake(env, dt):
    L, ops = env
    g = torch.Generator().manual_seed(3)
    M, N, K = 70, 96, 64
    x = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda() / 8
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    mask = (torch.rand(M, generator=g) > 0.3).float().cuda()
    alpha = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    xa, wp = cast(x, dt), ops.pack_linear(w, dt)
    out = torch.zeros(M, N, device="cuda")
    outa = torch.zeros(M, N, device="cuda", dtype=xa.dtype)
    ops.gemm(xa, wp, M, N, dtype=dt, lda=K, cin=K, bias=b, act="lrelu", slope=0.1, residual=res, ldr=N, rowmask=mask,
             alpha=alpha, out_f32=out, ldo_f=N, out_act=outa, ldo_a=N)
    v = (F.leaky_relu(F.linear(xa.float(), wp[:, :K].float(), b), 0.1) + res) * mask[:, None]
    assert rel_err(out, v) < TOL[dt]
    sn = v + (alpha + 1e-9).reciprocal() * torch.sin(alpha * v) ** 2
    assert rel_err(outa, sn) < (1e-2 if dt == 1 else 1e-4)


@pytest.mark.parametrize("dt", DTX)
@pytest.mark.parametrize("Cin,Cout,k,dil,T", [(80, 1536, 7, 1, 50), (48, 48, 7, 9, 700), (96, 96, 7, 3, 333), (192, 192, 1, 1, 100),
                                              (320, 256, 3, 1, 64)])
def test_conv1d(env, dt, Cin, Cout, k, dil, T):
    L, ops = env
    g = torch.Generator().manual_seed(Cin + k)
    B = 2
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = torch.randn(Cout, Cin, k, generator=g).cuda() / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g).cuda()
    xt = cast(x.transpose(1, 2).contiguous(), dt)
    wp = ops.pack_conv1d(w, dt)
    pad = (k - 1) * dil // 2 if k != 3 else 2              # k3 case = causal (left pad 2)
    out = torch.zeros(B, T, Cout, device="cuda")
    ops.conv1d(xt, wp, T=T, Cin=Cin, k=k, dil=dil, pad_left=pad, dtype=dt, batch=B, bias=b, out_f32=out)
    wq = wp[:, :k * Cin].float().reshape(Cout, k, Cin).permute(0, 2, 1)
    xin = xt.float().transpose(1, 2)
    if k == 3:
        ref = F.conv1d(F.pad(xin, (2, 0)), wq, b)
    else:
        ref = F.conv1d(xin, wq, b, dilation=dil, padding=pad)
    assert rel_err(out, ref.transpose(1, 2)) < TOL[dt]


@pytest.mark.parametrize("dt", DTX)
@pytest.mark.parametrize("Cin,Cout,s,T", [(1536, 768, 5, 8), (96, 48, 2, 40), (192, 96, 3, 33), (384, 192, 4, 17)])
def test_convtranspose1d(env, dt, Cin, Cout, s, T):
    L, ops = env
    g = torch.Generator().manual_seed(Cin + s)
    B = 2
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = torch.randn(Cin, Cout, 2 * s, generator=g).cuda() / math.sqrt(Cin * 2)
    b = torch.randn(Cout, generator=g).cuda()
    xt = cast(x.transpose(1, 2).contiguous(), dt)
    wq = w.to(L.WEIGHT_DT[dt]).float()
    wp = ops.pack_convtranspose1d(w, s, dt)
    out = torch.full((B, T * s, Cout), float("nan"), device="cuda")
    ops.convtranspose1d(xt, wp, T=T, Cin=Cin, Cout=Cout, stride=s, dtype=dt, batch=B, bias=b, out_f32=out)
    ref = F.conv_transpose1d(xt.float().transpose(1, 2), wq, b, stride=s, padding=math.ceil(s / 2),
                             output_padding=s % 2)
    assert ref.shape[-1] == T * s
    assert not torch.isnan(out).any()
    assert rel_err(out, ref.transpose(1, 2)) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C,rms", [(256, False), (512, False), (896, True), (80, False)])
def test_rownorm(env, dt, C, rms):
    L, ops = env
    g = torch.Generator().manual_seed(C)
    B, T = 2, 37
    x = (torch.randn(B, T, C, generator=g) * 2 + 0.5).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    add = torch.randn(B, C, generator=g).cuda()
    mask = (torch.rand(B, T, generator=g) > 0.2).float().cuda()
    outf = torch.zeros(B, T, C, device="cuda")
    outa = torch.zeros(B, T, C, device="cuda", dtype=torch.bfloat16 if dt else torch.float32)
    ops.rownorm(x, gamma, None if rms else beta, 1e-5, rows=T, C_=C, batch=B, rms=rms, act="none" if rms else "mish",
                rowmask=mask, addvec=add, out_f32=outf, out_act=outa, dtype=dt)
    if rms:
        y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma
    else:
        y = F.mish(F.layer_norm(x, (C,), gamma, beta, 1e-5))
    y = (y * mask[..., None] + add[:, None, :]) * mask[..., None]
    assert rel_err(outf, y) < 1e-5
    assert rel_err(outa, y) < (1e-2 if dt else 1e-5)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("T,chunk,relpos", [(50, 0, False), (130, 50, False), (77, 0, True), (100, 50, True)])
def test_attn_dense(env, dt, T, chunk, relpos):
    from oracle import flow as OF
    L, ops = env
    g = torch.Generator().manual_seed(T)
    B, H, D = 2, 8, 64
    q, k, v = (cast(torch.randn(B, T, H * D, generator=g).cuda(), dt) for _ in range(3))
    km = torch.ones(B, T)
    km[1, T - 9:] = 0
    out = torch.zeros(B, T, H * D, device="cuda", dtype=q.dtype)
    pos = pu = pv = None
    if relpos:
        pos = cast(torch.randn(2 * T - 1, H * D, generator=g).cuda(), dt)
        pu, pv = torch.randn(H, D, generator=g).cuda() * 0.2, torch.randn(H, D, generator=g).cuda() * 0.2
    ops.attn_dense(q, k, v, out, B=B, H=H, Tq=T, Tk=T, ldq=H * D, ldk=H * D, ldv=H * D, ldo=H * D, q_bs=T * H * D,
                   k_bs=T * H * D, v_bs=T * H * D, o_bs=T * H * D, scale=D ** -0.5, dtype=dt, keymask=km.cuda(),
                   chunk=chunk, pos=pos, ldp=H * D, pos_u=pu, pos_v=pv)
    # reference
    qh = q.float().cpu().view(B, T, H, D)
    kh = k.float().cpu().view(B, T, H, D).transpose(1, 2)
    vh = v.float().cpu().view(B, T, H, D).transpose(1, 2)
    if relpos:
        ph = pos.float().cpu().view(1, 2 * T - 1, H, D).transpose(1, 2)
        ac = (qh + pu.cpu()).transpose(1, 2) @ kh.transpose(-2, -1)
        bd = OF.rel_shift((qh + pv.cpu()).transpose(1, 2) @ ph.transpose(-2, -1))
        s = (ac + bd) * D ** -0.5
    else:
        s = (qh.transpose(1, 2) @ kh.transpose(-2, -1)) * D ** -0.5
    vis = km.bool()[:, None, :].expand(B, T, T).clone()
    if chunk:
        vis = vis & OF.subsequent_chunk_mask(T, chunk)[None]
    s = s.masked_fill(~vis[:, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, H * D)
    assert rel_err(out.cpu(), ref) < (2e-2 if dt else 2e-5)


@pytest.mark.parametrize("T,chunk,q_begin", [(64, 0, 0), (500, 0, 0), (137, 50, 0), (1000, 0, 0), (1000, 50, 944), (1450, 50, 1392),
                                              (600, 50, 544), (2999, 0, 2960)])
@pytest.mark.parametrize("fp8", [False, True])
def test_attn_flash_bf16(env, T, chunk, q_begin, fp8):
    """Every variant of the MFMA flash kernel: 32- and 16-query fragments per wave, and (q_begin > 0, many keys) the
    split-key variant a streaming hop uses; rows before q_begin must stay untouched."""
    from oracle import flow as OF
    L, ops = env
    g = torch.Generator().manual_seed(T + 1)
    B, H, D = 2, 8, 64
    q, k, v = (torch.randn(B, T, H * D, generator=g).cuda().bfloat16() for _ in range(3))
    Tp = ops.round_up(T, 8)
    vt = torch.zeros(B, H * D, Tp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :T] = v.transpose(1, 2)
    km = torch.ones(B, T)
    km[1, T - 9:] = 0
    out = torch.zeros(B, T, H * D, device="cuda", dtype=torch.bfloat16)
    ops.attn_flash_bf16(q, k, vt, out, B=B, H=H, T=T, ldq=H * D, ldk=H * D, ldvt=Tp, ldo=H * D, q_bs=T * H * D,
                        k_bs=T * H * D, vt_bs=H * D * Tp, o_bs=T * H * D, scale=D ** -0.5, keymask=km.cuda(), chunk=chunk,
                        q_begin=q_begin, fp8=fp8)
    qh, kh, vh = (t.float().cpu().view(B, T, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-2, -1)) * D ** -0.5
    vis = km.bool()[:, None, :].expand(B, T, T).clone()
    if chunk:
        vis = vis & OF.subsequent_chunk_mask(T, chunk)[None]
    s = s.masked_fill(~vis[:, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, H * D)
    err = rel_err(out.cpu()[:, q_begin:], ref[:, q_begin:])
    # fp8: e4m3 Q / K / V / P (3-bit mantissas): requirement 15 % of max |ref| at the worst element, 7 % of the RMS overall
    # (measured 4-10 % and 4.8-5.5 %: four operands with 3-bit mantissas, ~3.6 % RMS rounding each)
    assert err < (1.5e-1 if fp8 else 2e-2), err
    if fp8:
        d = out.cpu()[:, q_begin:].float() - ref[:, q_begin:]
        rms = float(d.pow(2).mean().sqrt() / ref[:, q_begin:].pow(2).mean().sqrt())
        print(f"fp8 flash T={T} chunk={chunk} q_begin={q_begin}: max abs err / max |ref| = {err:.3e}, rms err / rms ref = {rms:.3e}")
        assert rms < 7e-2, rms
    assert float(out[:, :q_begin].abs().max() if q_begin else 0.0) == 0.0


@pytest.mark.parametrize("T,chunk", [(16, 0), (77, 0), (100, 50), (500, 0), (1000, 100), (1003, 0)])
def test_attn_relpos_bf16(env, T, chunk):
    """Conformer rel-pos attention on the MFMA (mmx_attn_relpos_bf16) against the float64 statement of
    RelPositionMultiHeadedAttention (attention.py:215-330): (q + u) k^T + rel_shift((q + v) p^T), chunk mask, softmax, P V.
    Ragged lengths (last query fragment partly empty, window rows clamped at both ends), batch 2."""
    from oracle import flow as OF
    L, ops = env
    g = torch.Generator().manual_seed(T + 3)
    B, H, D = 2, 8, 64
    q, k, v = (torch.randn(B, T, H * D, generator=g).cuda().bfloat16() for _ in range(3))
    pos = torch.randn(2 * T - 1, H * D, generator=g).cuda().bfloat16()
    pu, pv = torch.randn(H, D, generator=g).cuda() * 0.2, torch.randn(H, D, generator=g).cuda() * 0.2
    Tp = ops.round_up(T, 8)
    vt = torch.zeros(B, H * D, Tp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :T] = v.transpose(1, 2)
    out = torch.zeros(B, T, H * D, device="cuda", dtype=torch.bfloat16)
    ops.attn_relpos_bf16(q, k, vt, pos, pu, pv, out, B=B, H=H, T=T, ldq=H * D, ldk=H * D, ldvt=Tp, ldp=H * D, ldo=H * D,
                         q_bs=T * H * D, k_bs=T * H * D, vt_bs=H * D * Tp, o_bs=T * H * D, scale=D ** -0.5, chunk=chunk)
    qh = q.double().cpu().view(B, T, H, D)
    kh, vh = (t.double().cpu().view(B, T, H, D).transpose(1, 2) for t in (k, v))
    ph = pos.double().cpu().view(1, 2 * T - 1, H, D).transpose(1, 2)
    ac = (qh + pu.double().cpu()).transpose(1, 2) @ kh.transpose(-2, -1)
    bd = OF.rel_shift((qh + pv.double().cpu()).transpose(1, 2) @ ph.transpose(-2, -1))
    s = (ac + bd) * D ** -0.5
    if chunk:
        s = s.masked_fill(~OF.subsequent_chunk_mask(T, chunk)[None, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, H * D)
    err = rel_err(out.cpu().double(), ref)
    assert err < 2e-2, err                              # bf16 operands (q + u, q + v, P rounded to bf16), as the flash kernel


@pytest.mark.parametrize("T,chunk", [(200, 0), (500, 0), (500, 50), (1000, 0), (130, 0)])
def test_attn_flash_klen(env, T, chunk):
    """klen (valid keys per batch row of a padded batch) gives the result of the same prefix mask passed as keymask, on
    every valid query row; padding rows come out finite (zeros where a whole workgroup is padding)."""
    from oracle import flow as OF
    L, ops = env
    g = torch.Generator().manual_seed(T)
    B, H, D = 3, 8, 64
    q, k, v = (torch.randn(B, T, H * D, generator=g).cuda().bfloat16() for _ in range(3))
    Tp = ops.round_up(T, 8)
    vt = torch.zeros(B, H * D, Tp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :T] = v.transpose(1, 2)
    lens = [T, T - 9, T // 2 + 5]
    km = torch.zeros(B, T)
    for b in range(B):
        km[b, :lens[b]] = 1
    klen = torch.tensor(lens, dtype=torch.int32).cuda()
    outs = []
    for kw in (dict(keymask=km.cuda()), dict(klen=klen)):
        out = torch.full((B, T, H * D), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.attn_flash_bf16(q, k, vt, out, B=B, H=H, T=T, ldq=H * D, ldk=H * D, ldvt=Tp, ldo=H * D, q_bs=T * H * D,
                            k_bs=T * H * D, vt_bs=H * D * Tp, o_bs=T * H * D, scale=D ** -0.5, chunk=chunk, **kw)
        outs.append(out.float().cpu())
    qh, kh, vh = (t.float().cpu().view(B, T, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-2, -1)) * D ** -0.5
    vis = km.bool()[:, None, :].expand(B, T, T).clone()
    if chunk:
        vis = vis & OF.subsequent_chunk_mask(T, chunk)[None]
    ref = (torch.softmax(s.masked_fill(~vis[:, None], float("-inf")), -1) @ vh).transpose(1, 2).reshape(B, T, H * D)
    for b in range(B):
        n = lens[b]
        assert rel_err(outs[1][b, :n], ref[b, :n]) < 2e-2, b
        assert rel_err(outs[1][b, :n], outs[0][b, :n]) < 1e-2, b        # same tiles, same order: bf16 rounding of P only
        assert torch.isfinite(outs[1][b]).all()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,K,N,epi,rs", [(1, 896, 1152, 0, True), (3, 896, 896, 2, False), (17, 4864, 896, 2, False),
                                          (1, 896, 4864, 1, True), (33, 896, 4864, 1, True), (2, 896, 6564, 0, True)])
def test_skinny_gemm(env, dt, B, K, N, epi, rs):
    L, ops = env
    g = torch.Generator().manual_seed(B * 31 + N)
    x = torch.randn(B, K, generator=g).cuda() * 3
    eps = 1e-6
    if epi == 1:
        w = torch.randn(2 * N, K, generator=g).cuda() / math.sqrt(K)
    else:
        w = torch.randn(N, K, generator=g).cuda() / math.sqrt(K)
    ks = (1 + 0.1 * torch.randn(K, generator=g)).cuda() if rs else None
    bias = torch.randn(N, generator=g).cuda() if epi == 0 else None
    wc = cast(w, dt).contiguous()
    wp = ops.pack_skinny(wc, dtype=dt, kscale=ks, interleave_half=(N if epi == 1 else 0))
    xin = x if (epi != 2) else cast(x, dt)                  # o_proj / down take the compute-dtype activation
    weff = cast(wc.float() * (ks if rs else 1.0), dt).float()
    xe = cast(xin, dt).float()
    acc = xe @ weff.t()
    if rs:
        acc = acc * torch.rsqrt(xin.float().pow(2).mean(-1, keepdim=True) + eps)
    if epi == 0:
        out = torch.zeros(B, N, device="cuda")
        ops.skinny_gemm(xin, wp, B=B, K=K, N=N, dtype=dt, bias=bias, rs=rs, eps=eps, epi=0, out_f32=out)
        ref = acc + bias
    elif epi == 1:
        out = torch.zeros(B, N, device="cuda", dtype=torch.bfloat16 if dt else torch.float32)
        ops.skinny_gemm(xin, wp, B=B, K=K, N=N, dtype=dt, rs=rs, eps=eps, epi=1, out_act=out)
        ref = F.silu(acc[:, :N]) * acc[:, N:]
    else:
        res = torch.randn(B, N, generator=g).cuda()
        out = res.clone()
        ops.skinny_gemm(xin, wp, B=B, K=K, N=N, dtype=dt, epi=2, out_f32=out)
        ref = res + acc
    assert rel_err(out, ref) < (1.5e-2 if dt else 3e-5)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,K,N,epi", [(9, 896, 1152, 0), (32, 896, 896, 2), (17, 4864, 896, 2), (32, 896, 4864, 1),
                                       (64, 896, 4864, 1), (33, 896, 6564, 0)])
def test_skinny_gemm_packed_activations(env, dt, B, K, N, epi):
    """MMX_X_PACKED / MMX_OUT_PACKED (MFMA-fragment-ordered activations, batch > 8): bit-identical to the row-major
    launch — the layout changes how x is fetched and where out_act lands, not the arithmetic."""
    L, ops = env
    g = torch.Generator().manual_seed(B * 13 + N)
    rs = epi != 2
    x = cast((torch.randn(B, K, generator=g) * 2).cuda(), dt).contiguous()
    w = (torch.randn((2 * N if epi == 1 else N), K, generator=g) / math.sqrt(K)).cuda()
    wp = ops.pack_skinny(cast(w, dt).contiguous(), dtype=dt, interleave_half=(N if epi == 1 else 0))
    res = torch.randn(B, N, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda() if epi == 0 else None
    tdt = torch.bfloat16 if dt else torch.float32
    outs = []
    for pk in (False, True):
        opk = pk and N % 32 == 0
        of = res.clone() if epi != 1 else None
        oa = torch.full((ops.packed_rows(B) * N,), float("nan"), device="cuda", dtype=tdt)
        ops.skinny_gemm(ops.pack_act(x, dt) if pk else x, wp, B=B, K=K, N=N, dtype=dt, bias=bias, rs=rs, eps=1e-6, epi=epi,
                        out_f32=of, out_act=oa, x_packed=pk, out_packed=opk)
        oa = ops.unpack_act(oa, B, N, dt) if opk else oa[:B * N].reshape(B, N)
        outs.append((of.clone() if of is not None else None, oa.float().clone()))
    if outs[0][0] is not None:
        assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.isfinite(outs[1][1]).all()


def test_pack_act_roundtrip_and_layout(env):
    L, ops = env
    for dt, E in ((1, 8), (0, 4)):
        x = torch.arange(20 * 64, dtype=torch.float32).reshape(20, 64)
        xp = ops.pack_act(x, dt)
        assert torch.equal(ops.unpack_act(xp, 20, 64, dt), x)
        KB = 4 * E
        row, col = 17, 45                                   # include/mmx_hip.h: xp[m][kb][g*16 + l16][j]
        idx = ((((row // 16) * (64 // KB) + col // KB) * 64) + ((col % KB) // E) * 16 + row % 16) * E + col % E
        assert xp[idx] == x[row, col]


@pytest.mark.parametrize("dt", DT)
def test_rope_kv_paged_attention_matches_oracle(env, dt):
    """prefill of 11 tokens + 3 decode steps through rope_kv_store / paged_attn vs the oracle's Qwen2 attention."""
    from oracle import llm as OL
    L, ops = env
    g = torch.Generator().manual_seed(5)
    Hq, Hkv, D, page = 14, 2, 64, 16
    B = 2
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).cuda()
    npages, maxp = 16, 4
    tdt = torch.bfloat16 if dt else torch.float32
    kc = torch.zeros(npages, Hkv, page, D, device="cuda", dtype=tdt)
    vc = torch.zeros_like(kc)
    bt = torch.tensor([[3, 7, 1, 0], [9, 2, 5, 4]], dtype=torch.int32).cuda()
    lens = [11, 6]
    hist = [[], []]
    for step, rows in enumerate([None, 1, 1, 1]):
        for b in range(B):
            r = lens[b] if rows is None else 1
            qkv = torch.randn(1, r, (Hq + 2 * Hkv) * D, generator=g)
            hist[b].append(qkv)
        pos = torch.tensor([sum(x.shape[1] for x in hist[b][:-1]) for b in range(B)], dtype=torch.int32).cuda()
        for b in range(B):   # variable rows per sequence in prefill: one call per sequence
            qkv = hist[b][-1].cuda().contiguous()
            r = qkv.shape[1]
            qo = torch.zeros(1, r, Hq * D, device="cuda", dtype=tdt)
            ops.rope_kv_store(qkv, inv, pos[b:b + 1], qo, kc, vc, bt[b:b + 1], B=1, rows=r, Hq=Hq, Hkv=Hkv, page=page, dtype=dt)
            out = torch.zeros(1, r, Hq * D, device="cuda", dtype=tdt)
            ops.paged_attn(qo, pos[b:b + 1], kc, vc, bt[b:b + 1], out, B=1, rows=r, Hq=Hq, Hkv=Hkv, page=page, dtype=dt)
            # oracle
            allq = torch.cat(hist[b], dim=1)
            n = allq.shape[1]
            qq = allq[..., :Hq * D].view(1, n, Hq, D).transpose(1, 2)
            kk = allq[..., Hq * D:(Hq + Hkv) * D].view(1, n, Hkv, D).transpose(1, 2)
            vv = allq[..., (Hq + Hkv) * D:].view(1, n, Hkv, D).transpose(1, 2)
            cos, sin = OL.rope_cos_sin(torch.arange(n), D, 1e6)
            qq = qq * cos + OL.rotate_half(qq) * sin
            kk = kk * cos + OL.rotate_half(kk) * sin
            kr, vr = kk.repeat_interleave(Hq // Hkv, 1), vv.repeat_interleave(Hq // Hkv, 1)
            s = (qq @ kr.transpose(-2, -1)) * D ** -0.5
            s = s.masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), float("-inf"))
            ref = (torch.softmax(s, -1) @ vr).transpose(1, 2).reshape(1, n, Hq * D)[:, n - r:]
            assert rel_err(out.cpu(), ref) < (2e-2 if dt else 2e-5), (step, b)


@pytest.mark.parametrize("dt,per_head", [(0, False), (1, True), (1, False)])
@pytest.mark.parametrize("ctx", [0, 5, 15, 16, 63, 300, 511, 512, 700])
def test_decode_attn_one_token(env, dt, per_head, ctx):
    """mmx_decode_attn (RoPE of q and the new k, KV append, causal GQA attention of ONE new token over a paged cache) vs
    torch: the per-head kernel (both dtypes) and the GQA-shared MFMA kernel (bf16), row-major and packed output, context
    lengths around the page (16) and round (512 keys) boundaries, shuffled pages, sequences of different length."""
    from oracle import llm as OL
    L, ops = env
    g = torch.Generator().manual_seed(ctx + 3)
    Hq, Hkv, D, page = 14, 2, 64, 16
    B = 5
    maxp = 64
    tdt = torch.bfloat16 if dt else torch.float32
    lens = [ctx, max(ctx - 7, 0), ctx // 2, ctx, min(ctx + 9, maxp * page - 1)]
    perm = torch.randperm(B * maxp, generator=g).to(torch.int32)
    bt = perm.reshape(B, maxp).contiguous().cuda()
    kc = (torch.randn(B * maxp, Hkv, page, D, generator=g) * 0.7).to(tdt).cuda()
    vc = torch.randn(B * maxp, Hkv, page, D, generator=g).to(tdt).cuda()
    kc0, vc0 = kc.clone(), vc.clone()
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, generator=g).cuda()
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).cuda()
    pos = torch.tensor(lens, dtype=torch.int32).cuda()
    ang = torch.arange(maxp * page, dtype=torch.float32)[:, None] * inv.cpu()[None, :]
    tab = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous().cuda()          # as LlmEngine builds it
    for packed in (False, True):
        kc.copy_(kc0)
        vc.copy_(vc0)
        out = torch.zeros(ops.packed_rows(B), Hq * D, device="cuda", dtype=tdt)
        ops.decode_attn(qkv, inv, pos, kc, vc, bt, out, B=B, Hq=Hq, Hkv=Hkv, page=page, dtype=dt, out_packed=packed,
                        per_head=per_head, rope_tab=tab)
        got = ops.unpack_act(out, B, Hq * D, dt).float().cpu() if packed else out[:B].float().cpu()
        if dt == 0 or per_head:                            # two query heads per workgroup (the default) = one per workgroup, bit for bit
            kc1, vc1, out1 = kc0.clone(), vc0.clone(), torch.zeros_like(out)
            ops.decode_attn(qkv, inv, pos, kc1, vc1, bt, out1, B=B, Hq=Hq, Hkv=Hkv, page=page, dtype=dt, out_packed=packed,
                            per_head=per_head, rope_tab=tab, one_head=True)
            assert torch.equal(out1, out) and torch.equal(kc1, kc) and torch.equal(vc1, vc)
        for b in range(B):
            n = lens[b]
            rows = [(int(bt[b, j // page]), j % page) for j in range(n)]
            kk = torch.stack([kc0[pg, :, r] for pg, r in rows], 1).float().cpu() if n else torch.zeros(Hkv, 0, D)
            vv = torch.stack([vc0[pg, :, r] for pg, r in rows], 1).float().cpu() if n else torch.zeros(Hkv, 0, D)
            x = qkv[b].cpu()
            cos, sin = OL.rope_cos_sin(torch.tensor([n]), D, 1e6)
            q = x[:Hq * D].view(Hq, 1, D)
            kn = x[Hq * D:(Hq + Hkv) * D].view(Hkv, 1, D)
            vn = x[(Hq + Hkv) * D:].view(Hkv, 1, D)
            q = q * cos + OL.rotate_half(q) * sin
            kn = kn * cos + OL.rotate_half(kn) * sin
            kn, vn = kn.to(tdt).float(), vn.to(tdt).float()              # the cache holds them in the storage type
            kk, vv = torch.cat([kk, kn], 1), torch.cat([vv, vn], 1)
            kr, vr = kk.repeat_interleave(Hq // Hkv, 0), vv.repeat_interleave(Hq // Hkv, 0)
            s = (q @ kr.transpose(-2, -1)) * D ** -0.5
            ref = (torch.softmax(s, -1) @ vr).reshape(Hq * D)
            assert rel_err(got[b], ref) < (1e-2 if dt else 2e-5), (packed, b, n)
            pg, r = int(bt[b, n // page]), n % page
            assert rel_err(kc[pg, :, r].float().cpu(), kn[:, 0]) < (1e-2 if dt else 1e-6)       # the append landed
            assert torch.equal(vc[pg, :, r].float().cpu(), vn[:, 0])


def test_sampler_matches_oracle(env):
    """Device sampler (log_softmax + RAS + EOS re-draw + bookkeeping) vs oracle.llm.sampling_ids_e on the same
    Philox noise: ids must be IDENTICAL."""
    from oracle import llm as OL
    from oracle import weights as W
    L, ops = env
    cases = W.sampler_cases(64)
    V, E, eos, seed = 6564, 896, 6561, 1234
    emb = torch.randn(V, E).cuda()
    mism = 0
    for s, (logp, hist) in enumerate(cases):
        logits = (logp + 3.7).unsqueeze(0)                   # any shift: log_softmax removes it
        if s % 5 == 0:
            logits[0, eos] = logits.max() + 2.0              # make EOS dominant -> exercises the re-draw loop
        step = s
        min_len = step + 1 if s % 2 == 0 else 0              # ignore_eos on even cases
        state = torch.tensor([40, step, len(hist), 0, min_len, 999, 3, 0], dtype=torch.int32).cuda()   # field-major, B = 1
        out_tokens = torch.zeros(1, 64, dtype=torch.int32)
        out_tokens[0, :len(hist)] = torch.tensor(hist, dtype=torch.int32)
        out_tokens = out_tokens.cuda()
        sampled = torch.full((1, 64), -1, dtype=torch.int32).cuda()
        nx = torch.zeros(1, E).cuda()
        lp = torch.zeros(1, V).cuda()
        ops.sample_step(logits.cuda().contiguous(), state, out_tokens, emb, nx, V=V, B=1, eos_id=eos, seed=seed,
                        sampled=sampled, logp_out=lp)
        got = int(sampled[0, step])
        lpo = logits[0].log_softmax(-1)
        assert (lp[0].cpu() - lpo).abs().max() < 1e-4
        want = OL.sampling_ids_e(lpo, hist, lambda k: OL.philox_noise(seed, 3, step, k), ignore_eos=step < min_len, eos=eos)
        mism += got != want
        st = state.cpu().tolist()
        if want == eos:
            assert st[3] == 1
        elif want < eos:
            assert st[2] == len(hist) + 1 and int(out_tokens[0, len(hist)]) == want
            assert torch.equal(nx[0], emb[want])
            assert st[0] == 41 and st[1] == step + 1
    assert mism == 0, f"{mism} sampled ids differ from the oracle"


# ----------------------------------------------------------------------------- fused estimator row-tile kernels (csrc/fused.hip)
FUSED_BM = {0: [16, 32], 1: [16, 32, 64]}
FUSED_TOL = {0: 5e-5, 1: 3e-2}


def _pk(ops, w, dt):
    return ops.pack_skinny(cast(w, dt).contiguous(), dtype=dt)


def _ln(x, g, b):
    return F.layer_norm(x, x.shape[-1:], g, b, 1e-5)


def _check_qkv(ops, dt, hn, wqkv, qk, vt, B, T, tol, tag):
    ref = hn @ wqkv.t()
    if dt == 1:
        assert rel_err(qk[:, :T], ref[..., :1024]) < tol, (tag, "qk", rel_err(qk[:, :T], ref[..., :1024]))
        assert rel_err(vt[:, :, :T].transpose(1, 2), ref[..., 1024:]) < tol, (tag, "vt")
        assert float(vt[:, :, T:].abs().max()) == 0.0 if vt.shape[2] > T else True
    else:
        assert rel_err(qk[:, :T], ref) < tol, (tag, "qkv", rel_err(qk[:, :T], ref))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,masked,with_next", [(2, 50, False, True), (3, 37, True, False), (1, 130, True, True), (2, 64, False, True)])
def test_est_tail_fused(env, dt, B, T, masked, with_next):
    """attn-out projection + residual -> LayerNorm -> FF1 + GELU -> FF2 + residual (-> LayerNorm -> Q/K/V of the next
    block) in one launch vs plain torch fp32 (matcha transformer.py:286-313, 256-285)."""
    L, ops = env
    g = torch.Generator().manual_seed(100 * B + T)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    ao, x = rn(B, T, 512), rn(B, T, 256)
    wo, w1, w2, wqkv = rn(256, 512, sc=512 ** -0.5), rn(1024, 256, sc=1 / 16), rn(256, 1024, sc=1 / 32), rn(1536, 256, sc=1 / 16)
    bo, b1, b2 = rn(256, sc=0.1), rn(1024, sc=0.1), rn(256, sc=0.1)
    n3g, n3b, n1g, n1b = 1 + rn(256, sc=0.1), rn(256, sc=0.1), 1 + rn(256, sc=0.1), rn(256, sc=0.1)
    mask = (torch.rand(B, T, generator=g) > 0.3).float().cuda() if masked else None
    aoq, wq = cast(ao, dt), lambda w: cast(w, dt).float()
    x1 = x + aoq.float() @ wq(wo).t() + bo
    h = cast(_ln(x1, n3g, n3b), dt).float()
    ff = cast(F.gelu(h @ wq(w1).t() + b1), dt).float() @ wq(w2).t() + b2
    x2 = x1 + ff
    if mask is not None:
        x2 = x2 * mask[..., None]
    hn = cast(_ln(x2, n1g, n1b), dt).float()
    w = dict(wo_p=_pk(ops, wo, dt), w1_p=_pk(ops, w1, dt), w2_p=_pk(ops, w2, dt), bo=bo, b1=b1, b2=b2, n3g=n3g, n3b=n3b)
    wqkv_p = _pk(ops, wqkv, dt)
    tol = FUSED_TOL[dt]
    # every kernel variant of the build: the library defaults (bf16: 8 waves with 32-column passes for 64 / 32 rows), the
    # one-wave-per-SIMD kernels and the 64-column-pass 8-wave ones (explicit cfg)
    variants = [(bm, {}) for bm in FUSED_BM[dt]]
    if dt == 1:
        variants += [(64, dict(waves=4, pf=2)), (64, dict(waves=4, pf=4)), (32, dict(waves=4, pf=4)), (32, dict(waves=8, pf=2)),
                     (16, dict(waves=8)), (32, dict(narrow=True, pf=4))]
    for bm, cfg in variants:
        xio = x.clone()
        Tp = ops.round_up(T, 8)
        ldq = 1024 if dt == 1 else 1536
        qk = torch.zeros(B, T, ldq, device="cuda", dtype=L.TORCH_DT[dt])
        vt = torch.full((B, 512, Tp), 7.0, device="cuda", dtype=L.TORCH_DT[dt]) if dt == 1 else None
        act = torch.zeros(B, T, 512, device="cuda", dtype=L.TORCH_DT[dt])
        nxt = ops.est_next(wqkv=wqkv_p, n1g=n1g, n1b=n1b, q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp, vt_bs=512 * Tp) if with_next else None
        ops.est_tail(aoq.contiguous(), xio, w, B=B, T=T, dtype=dt, bm=bm, rowmask=mask, act_out=act[:, :, 256:], act_ld=512, nxt=nxt, **cfg)
        torch.cuda.synchronize()
        assert rel_err(xio, x2) < tol, (bm, "x", rel_err(xio, x2))
        assert rel_err(act[:, :, 256:], x2) < max(tol, 1e-2 if dt == 1 else 0) and float(act[:, :, :256].abs().max()) == 0.0
        if with_next:
            _check_qkv(ops, dt, hn, wq(wqkv), qk, vt, B, T, tol, bm)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,cin,lda,masked", [(2, 50, 320, 320, False), (2, 45, 256, 512, True), (1, 100, 512, 512, True), (3, 16, 256, 256, False)])
def test_est_resnet_fused(env, dt, B, T, cin, lda, masked):
    """CausalResnetBlock1D (+ LayerNorm + Q/K/V of the following block) in one launch vs plain torch fp32
    (flow/decoder.py:65-85, matcha decoder.py:56-61)."""
    L, ops = env
    g = torch.Generator().manual_seed(7 * B + T + cin)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    mask = (torch.rand(B, T, generator=g) > 0.25).float().cuda() if masked else None
    m3 = mask[..., None] if masked else 1.0
    a_full = rn(B, T, lda) * m3
    a_in = cast(a_full, dt)
    wc1, wc2, wcr = rn(256, cin, 3, sc=(3 * cin) ** -0.5), rn(256, 256, 3, sc=768 ** -0.5), rn(256, cin, 1, sc=cin ** -0.5)
    b1, b2, br = rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1)
    g1, be1, g2, be2 = 1 + rn(256, sc=0.1), rn(256, sc=0.1), 1 + rn(256, sc=0.1), rn(256, sc=0.1)
    tv_all = rn(B, 3 * 256)
    tv = tv_all[:, 256:512]
    n1g, n1b, wqkv = 1 + rn(256, sc=0.1), rn(256, sc=0.1), rn(1536, 256, sc=1 / 16)
    wq = lambda w: cast(w, dt).float()
    xin = a_in.float()[..., :cin].transpose(1, 2)                       # [B, cin, T]
    conv = lambda z, w, b: F.conv1d(F.pad(z, (w.shape[2] - 1, 0)), wq(w), b).transpose(1, 2)
    h = F.mish(_ln(conv(xin, wc1, b1), g1, be1)) * m3
    h = cast((h + tv[:, None, :]) * m3, dt).float()
    h = F.mish(_ln(conv(h.transpose(1, 2), wc2, b2), g2, be2)) * m3
    xo = h + conv(xin, wcr, br)
    hn = cast(_ln(xo, n1g, n1b), dt).float()
    pc = lambda w: ops.pack_skinny(ops.pack_conv1d(w, dt), dtype=dt)
    r = dict(w1_p=pc(wc1), w2_p=pc(wc2), wr_p=pc(wcr), b1=b1, g1=g1, be1=be1, b2=b2, g2=g2, be2=be2, br=br)
    wqkv_p = _pk(ops, wqkv, dt)
    tol = FUSED_TOL[dt]
    for bm in ([16] if dt == 0 else [16, 32, 64]):
        x = torch.zeros(B, T, 256, device="cuda")
        Tp = ops.round_up(T, 8)
        ldq = 1024 if dt == 1 else 1536
        qk = torch.zeros(B, T, ldq, device="cuda", dtype=L.TORCH_DT[dt])
        vt = torch.full((B, 512, Tp), 7.0, device="cuda", dtype=L.TORCH_DT[dt]) if dt == 1 else None
        nxt = ops.est_next(wqkv=wqkv_p, n1g=n1g, n1b=n1b, q_out=qk, ldq=ldq, q_bs=T * ldq, vt_out=vt, ldvt=Tp, vt_bs=512 * Tp)
        ops.est_resnet(a_in.contiguous(), lda, cin, x, r, tv, tv_all.shape[1], B=B, T=T, dtype=dt, bm=bm, rowmask=mask, nxt=nxt)
        torch.cuda.synchronize()
        assert rel_err(x, xo) < tol, (bm, "x", rel_err(x, xo))
        _check_qkv(ops, dt, hn, wq(wqkv), qk, vt, B, T, tol, bm)


# dtype 2 = the split build (two bf16 planes per LDS tile): 2^-17 per operand, bounded like the split golden tests
@pytest.mark.parametrize("dt,tol", [(0, 2e-4), (1, 8e-2), (2, 5e-4)])
@pytest.mark.parametrize("B,T,masked,streaming", [(2, 64, False, False), (2, 500, False, False), (4, 130, True, True), (16, 320, True, False)])
def test_estimator_fused_equals_unfused(dt, tol, B, T, masked, streaming):
    """The whole estimator on the row-tile kernels (every tile size the host picks: 16 / 32 / 64 rows) vs the
    one-launch-per-op composition, same weights."""
    from mmx import shapes, synth
    from mmx.flow import FlowEngine
    sd = synth.synth_state_dict(shapes.flow_manifest(num_mid_blocks=2), 3)
    ef, eu = (FlowEngine(sd, dtype=dt, parts=("estimator",), fused=f) for f in (True, False))
    g = torch.Generator().manual_seed(B + T)
    x, mu = torch.randn(B, 80, T, generator=g).cuda(), torch.randn(B, 80, T, generator=g).cuda()
    cond, spks = torch.randn(B, 80, T, generator=g).cuda() * 0.5, torch.randn(B, 80, generator=g).cuda()
    t = torch.rand(B, generator=g).cuda()
    mask = torch.ones(B, 1, T).cuda()
    if masked:
        for b in range(B):
            mask[b, :, T - 7 * b:] = 0
    a = ef.estimator_channels_first(x, mask, mu, t, spks, cond, streaming)
    b_ = eu.estimator_channels_first(x, mask, mu, t, spks, cond, streaming)
    err = (a - b_).abs().max().item()
    print(f"fused vs unfused estimator dtype {dt} B={B} T={T}: max abs diff {err:.3e} (std {b_.std().item():.3f})")
    assert torch.isfinite(a).all() and err < tol, err


@pytest.mark.parametrize("T,T2", [(500, 400), (37, 46), (8, 8), (3, 10)])
def test_resample_linear_matches_torch_interpolate(T, T2):
    """mmx_resample_linear (the speed change of cli/model.py:312-314) against torch's own F.interpolate(mode="linear") on the CPU."""
    import torch.nn.functional as F
    from mmx import ops
    x = torch.randn(2, 80, T, generator=torch.Generator().manual_seed(T))
    got = ops.resample_linear(x.cuda(), T2).cpu()
    ref = F.interpolate(x, size=T2, mode="linear")
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-6
