"""GPU micro-benchmarks of single kernels (HIP events around back-to-back launches on the current stream)."""
import os
import sys
import math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import torch
from mmx import ops, _lib as L


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def gemm_case(M, N, K, batch=1, dt=1, out="act", act="none"):
    tdt = torch.bfloat16 if dt else torch.float32
    x = torch.randn(batch, M, K, device="cuda").to(tdt)
    w = ops.pack_linear(torch.randn(N, K, device="cuda") / math.sqrt(K), dt)
    of = torch.empty(batch, M, N, device="cuda") if out in ("f32", "both") else None
    oa = torch.empty(batch, M, N, device="cuda", dtype=tdt) if out in ("act", "both") else None
    fn = lambda: ops.gemm(x, w, M, N, dtype=dt, lda=K, cin=K, batch=batch, a_bstride=M * K, act=act,
                          out_f32=of, ldo_f=N, of_bstride=M * N, out_act=oa, ldo_a=N, oa_bstride=M * N)
    us = timeit(fn)
    fl = 2.0 * M * N * K * batch
    print(f"gemm M={M:6d} N={N:5d} K={K:5d} b={batch:3d} out={out:5s} act={act:5s}: {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    if which == "gemm":
        for K in (32, 256, 1024):
            gemm_case(500, 1024, K, 2)
        gemm_case(500, 256, 512, 2, out="f32")
        gemm_case(500, 256, 1024, 2, out="f32")
        gemm_case(500, 1024, 256, 16)
        gemm_case(500, 1024, 256, 16, act="gelu")
        gemm_case(500, 256, 1024, 16, out="f32")
        gemm_case(8000, 1024, 256, 1)
        gemm_case(8192, 8192, 1024, 1)
        gemm_case(8192, 8192, 8192, 1)
        gemm_case(240000, 48, 336, 1, out="both")
        gemm_case(40000, 192, 1344, 1, out="both")
    if which == "launch":
        x = torch.zeros(64, 256, device="cuda")
        g = torch.ones(256, device="cuda")
        o = torch.empty(64, 256, device="cuda")
        print("rownorm 64x256:", timeit(lambda: ops.rownorm(x, g, g, 1e-5, rows=64, C_=256, out_f32=o, dtype=0)), "us")
        x = torch.zeros(2, 500, 256, device="cuda")
        o = torch.empty(2, 500, 256, device="cuda", dtype=torch.bfloat16)
        print("rownorm 2x500x256:", timeit(lambda: ops.rownorm(x, g, g, 1e-5, rows=500, C_=256, batch=2, out_act=o, dtype=1)), "us")
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(100):
                ops.rownorm(x, g, g, 1e-5, rows=500, C_=256, batch=2, out_act=o, dtype=1)
        print("rownorm 2x500x256 in a 100-node graph:", timeit(gr.replay, iters=20, warm=3) / 100, "us per node")


def graph_time(fn, reps=20, iters=10):
    """per-launch time of `fn` inside a hipGraph of `reps` back-to-back launches (no host launch floor)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return timeit(g.replay, iters=iters, warm=2) / reps


def flash_case(B, T, chunk=0, fp8=False):
    H, D = 8, 64
    q, k = (torch.randn(B, T, H * D, device="cuda").bfloat16() for _ in range(2))
    Tp = ops.round_up(T, 8)
    vt = torch.randn(B, H * D, Tp, device="cuda").bfloat16()
    out = torch.empty(B, T, H * D, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.attn_flash_bf16(q, k, vt, out, B=B, H=H, T=T, ldq=H * D, ldk=H * D, ldvt=Tp, ldo=H * D, q_bs=T * H * D,
                                     k_bs=T * H * D, vt_bs=H * D * Tp, o_bs=T * H * D, scale=0.125, chunk=chunk, fp8=fp8)
    us = graph_time(fn)
    fl = 4.0 * B * H * T * T * D
    print(f"flash{' fp8' if fp8 else '    '} B={B:3d} T={T:5d} chunk={chunk}: {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "flash":
    for B, T in ((2, 128), (2, 512), (2, 1024), (2, 3000), (16, 512), (16, 1024), (64, 512)):
        flash_case(B, T)
        flash_case(B, T, fp8=True)
    flash_case(16, 512, 50)


def gemm_graph_case(M, N, K, batch, act="none", out="act", tile="auto"):
    dt = 1
    x = torch.randn(batch, M, K, device="cuda").bfloat16()
    w = ops.pack_linear(torch.randn(N, K, device="cuda") / math.sqrt(K), dt)
    of = torch.empty(batch, M, N, device="cuda") if out in ("f32", "both") else None
    oa = torch.empty(batch, M, N, device="cuda", dtype=torch.bfloat16) if out in ("act", "both") else None
    res = torch.randn(batch, M, N, device="cuda") if out == "f32" else None
    fn = lambda: ops.gemm(x, w, M, N, dtype=dt, lda=K, cin=K, batch=batch, a_bstride=M * K, act=act, residual=res, ldr=N,
                          r_bstride=M * N, out_f32=of, ldo_f=N, of_bstride=M * N, out_act=oa, ldo_a=N, oa_bstride=M * N,
                          tile=L.TILES[tile])
    us = graph_time(fn)
    print(f"tile={tile:8s} M={M} N={N:5d} K={K:5d} b={batch:3d} act={act:5s} out={out}: {us:7.2f} us {2.0*M*N*K*batch/us/1e6:7.1f} TF")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tiles":
    for (N, K, act, out) in ((1024, 256, "none", "act"), (1024, 256, "gelu", "act"), (256, 1024, "none", "f32"), (256, 512, "none", "f32"), (256, 768, "none", "f32")):
        for tile in ("auto", "128x128", "128x64", "64x64", "32x64"):
            gemm_graph_case(512, N, K, 16, act, out, tile)
            gemm_graph_case(512, N, K, 2, act, out, tile)


def skinny_case(B, K, N, epi, packed=False):
    w = (torch.randn((2 * N if epi == 1 else N), K, device="cuda") / 30).bfloat16()
    ws = [ops.pack_skinny(w, dtype=1, interleave_half=(N if epi == 1 else 0)) for _ in range(12)]
    x = torch.randn(B, K, device="cuda").bfloat16()
    if packed:
        x = ops.pack_act(x, 1)
    outf = torch.zeros(B, N, device="cuda") if epi != 1 else None
    outa = torch.empty(ops.packed_rows(B), N, device="cuda", dtype=torch.bfloat16)
    it = [0]

    def fn():
        it[0] += 1
        ops.skinny_gemm(x, ws[it[0] % 12], B=B, K=K, N=N, dtype=1, rs=(epi != 2), epi=epi, out_f32=outf, out_act=outa,
                        x_packed=packed, out_packed=packed and N % 32 == 0)
    us = graph_time(fn, reps=48)
    nbytes = w.numel() * 2
    print(f"skinny B={B:3d} K={K:5d} N={N:5d} epi={epi} packed={int(packed)}: {us:7.2f} us  {nbytes / us / 1e3:8.1f} GB/s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "skinny":
    for B in (1, 8, 16, 17, 32, 48, 64):
        skinny_case(B, 896, 4864, 1)
    for B in (1, 16, 32):
        skinny_case(B, 896, 1152, 0)
        skinny_case(B, 896, 896, 2)
        skinny_case(B, 4864, 896, 2)
    for B in (4, 8, 12, 16, 32, 64):
        for pk in (False, True):
            skinny_case(B, 896, 4864, 1, pk)
            skinny_case(B, 896, 1152, 0, pk)
            skinny_case(B, 896, 896, 2, pk)
            skinny_case(B, 4864, 896, 2, pk)


def decode_attn_case(B, ctx, layers=6, per_head=False):
    """mmx_decode_attn (RoPE + KV append + GQA attention of one new token) at context length ctx, rotating over layers."""
    Hq, Hkv, D, page = 14, 2, 64, 16
    max_pages = 2048 // page
    kc = (torch.randn(layers, B * max_pages + 1, Hkv, page, D, device="cuda") * 0.5).bfloat16()
    vc = torch.randn_like(kc)
    bt = torch.arange(B * max_pages, dtype=torch.int32, device="cuda").reshape(B, max_pages).contiguous()
    pos = torch.full((B,), ctx, dtype=torch.int32, device="cuda")
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda")
    inv_freq = (1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))).cuda()
    ang = torch.arange(2048, dtype=torch.float32)[:, None] * inv_freq.cpu()[None, :]
    tab = torch.cat([ang.cos(), ang.sin()], dim=1).contiguous().cuda()
    out = torch.empty(ops.packed_rows(B), Hq * D, device="cuda", dtype=torch.bfloat16)
    it = [0]

    def fn():
        it[0] += 1
        l = it[0] % layers
        ops.decode_attn(qkv, inv_freq, pos, kc[l], vc[l], bt, out, B=B, Hq=Hq, Hkv=Hkv, page=page, dtype=1, rope_tab=tab,
                        out_packed=B >= 4, per_head=per_head)
    us = graph_time(fn, reps=48)
    print(f"decode_attn {'per head  ' if per_head else 'GQA shared'} B={B:3d} ctx={ctx:5d}: {us:7.2f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "decode_attn":
    for B in (1, 16, 32):
        for ctx in (64, 300, 600, 1500):
            for ph in (True, False):
                decode_attn_case(B, ctx, per_head=ph)


def sampler_case(B, scale):
    """mmx_sample_step on random logits (std `scale`: 3 = the peaked synthetic LM, 0.5 = nearly flat)."""
    V, H = 6564, 896
    logits = torch.randn(B, V, device="cuda") * scale
    emb = torch.randn(V, H, device="cuda")
    x = torch.zeros(B, H, device="cuda")
    out = torch.zeros(B, 4096, dtype=torch.int32, device="cuda")
    st0 = torch.zeros(8, B, dtype=torch.int32, device="cuda")
    st0[4] = 1 << 20                                     # min_len: never stop on eos
    st0[5] = 1 << 20
    st = st0.clone()

    def fn():
        ops.sample_step(logits, st, out, emb, x, V=V, B=B, eos_id=6561, seed=1)
    us = graph_time(fn, reps=40)
    print(f"sampler B={B:3d} logit std {scale}: {us:7.2f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sampler":
    for B in (1, 32):
        for sc in (3.0, 0.5):
            sampler_case(B, sc)
