"""Where a decode-step projection kernel (csrc/decode.hip) spends its time: shader-clock stamps of the kernel's phases per wave
(the lab build's mmx_lab_skinny_stamps switch: run with MMX_LIB=minimax-speech_amd/lib/libmmx_hip_lab.so, `make -C minimax-speech_amd/csrc lab`), for the four projection shapes of a layer at batch 32, each launched back to back over 24
different weight sets (so the weights come from HBM as in the decode step).  Also the floor of a dependent launch chain: the
same number of trivial kernels in one hipGraph.

    python tools/decode_lab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimax-speech_amd"))
import ctypes as C  # noqa: E402

from mmx import _lib, ops  # noqa: E402

X3 = 3
LIB = _lib.load()
assert hasattr(LIB, "mmx_lab_skinny_stamps"), "run with MMX_LIB=<repo>/minimax-speech_amd/lib/libmmx_hip_lab.so (make -C minimax-speech_amd/csrc lab)"


def main():
    dev = "cuda"
    B, H, I, NQ = 32, 896, 4864, 1152
    g = torch.Generator().manual_seed(0)
    L = 24
    rnd = lambda n, k: (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    shapes = {"qkv": (H, NQ, 0, 1, 1, True), "o": (H, H, 2, 1, 1, False), "gu": (H, I, 1, 2, 1, True), "down": (I, H, 2, 2, 8, False)}
    for name, (K, N, epi, tw, J, rs) in shapes.items():
        ws = [ops.pack_skinny(rnd(2 * N if epi == 1 else N, K), dtype=X3, interleave_half=(N if epi == 1 else 0)) for _ in range(L)]
        x = torch.randn(B, K, generator=g).to(dev)
        xs = ops.split_planes(x)
        ssq = torch.zeros(32, 64, device=dev)
        ssq[:, :K // 16] = 1.0
        out = torch.zeros(B, N, device=dev)
        xs_out = torch.zeros(3, ops.plane_elems(B, N), dtype=torch.bfloat16, device=dev) if N % 32 == 0 else None
        ssq_out = torch.zeros(32, 64, device=dev)
        nt = (N + 15) // 16
        part = torch.zeros(J * nt * 8 * 64, device=dev) if J > 1 else None
        tickets = torch.zeros(nt, dtype=torch.int32, device=dev) if J > 1 else None
        nwg = ((nt + tw - 1) // tw) * J
        stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)

        def run(l, st=None):
            # lab build only (include/mmx_hip_lab.h): the stamp buffer is a device global of libmmx_hip_lab.so
            assert LIB.mmx_lab_skinny_stamps(C.c_void_p(st.data_ptr() if st is not None else 0)) == 0
            ops.skinny2(xs, ws[l], B=B, K=K, N=N, dtype=X3, ssq_in=(ssq if rs else None), epi=epi, out=(out if epi != 1 else None),
                        xs_out=(xs_out if epi != 0 else None), gamma_next=None, ssq_out=(ssq_out if epi == 2 else None),
                        tiles_per_wg=tw, ksplit=J, part=part, tickets=tickets)

        for l in range(L):
            run(l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for r in range(4):
                for l in range(L):
                    run(l)
        gr.replay()
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (4 * L)
        # one stamped launch after 23 unstamped ones (caches in the steady state of the loop)
        for l in range(L - 1):
            run(l)
        run(L - 1, stamps)
        torch.cuda.synchronize()
        st = stamps.cpu().reshape(nwg, 8, 8).double()
        # stamps are per-XCD counters: only differences inside one wave mean anything.  Slot 7 = kernel entry of the wave.
        order = [(7, "entry"), (0, "arguments in, indices computed"), (1, "loads issued"), (2, "loads landed"), (3, "mfma done, partials in LDS"),
                 (4, "after the barrier"), (5, "ticket taken"), (6, "end")]
        print(f"{name}: K={K} N={N} epi={epi} tw={tw} J={J}: {nwg} workgroups, {us:.2f} us per launch (hipGraph, back to back)")
        prev = 7
        for slot, nm in order[1:]:
            v, p0 = st[:, :, slot], st[:, :, prev]
            ok = (v > 0) & (p0 > 0)
            if not ok.any():
                continue
            d = (v - p0)[ok]
            tot = (v - st[:, :, 7])[ok]
            print(f"    {nm:32s} +{d.median():7.0f} ticks (median; max +{d.max():7.0f})   since entry: median {tot.median():7.0f} max {tot.max():7.0f}")
            prev = slot
    # floor of a dependent chain of trivial launches
    y = torch.zeros(B, H, device=dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(122):
            y.add_(1.0)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    e1.synchronize()
    print(f"122 dependent trivial launches in one hipGraph: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per replay "
          f"({e0.elapsed_time(e1) * 1e3 / 20 / 122:.2f} us per launch)")
    # (tick length of s_memtime: compare the "end" stamp's maximum with the launch time measured by HIP events)


if __name__ == "__main__":
    main()
